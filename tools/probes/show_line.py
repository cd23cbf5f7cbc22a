import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
print({k: (v["ms_per_step"], v["launches"]) for k, v in list(d["families"].items())[:8]})
print(d.get("with_input_feed")); print({k: (v["value"]) for k, v in d.get("extra_configs", {}).items()})
