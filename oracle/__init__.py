"""CPU oracle for the STrajNet hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (strajnet_amd) never imports it and fails
loudly when its HIP library is missing.

PARITY UNPINNED: the reference is pure Python on TensorFlow / tensorflow_addons,
neither of which exists in this image (and the reference ships no tests, golden
vectors or fixtures).  The oracle is therefore a literal restatement of the
reference source (file:line cited per function) pinned only by
  * hand-derivable known-answer tests (tests/test_oracle_kat.py), and
  * a second, independently formulated restatement (oracle/torch_ref.py)
    agreeing with oracle/np_ref.py to ~1e-10 in float64.
"""
