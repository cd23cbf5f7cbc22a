#!/usr/bin/env python
"""Dump the captured train-step hipGraph (hipGraphDebugDotPrint through torch's debug mode) and list the predecessors of chosen kernels:
which edges does the capture REALLY contain in front of the agent branch?   usage: tools/probes/graph_edges.py [substring ...]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss
from strajnet_amd.loss import OccupancyFlowTaskConfig
from strajnet_amd.graph import GraphedTrainStep

_G = torch.cuda.CUDAGraph
torch.cuda.CUDAGraph = lambda *a, **k: _G(keep_graph=True)          # keep the captured hipGraph_t so that it can be printed
dev = torch.device('cuda:0')
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), ogm_weight=1000.0, occ_weight=1000.0, flow_weight=1.0, replica=1.0,
                       flow_origin_weight=1000.0, no_use_warp=False, use_pred=False, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev, 256)
g = GraphedTrainStep(model, loss_fn, x)
path = '/tmp/step_graph.dot'
g.graph.debug_dump(path)
txt = open(path).read()
print('dot bytes', len(txt))
nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
pred = {}
for a, b in edges:
    pred.setdefault(b, []).append(a)
print(len(nodes), 'nodes', len(edges), 'edges')
for pat in (sys.argv[1:] or ['agent_prep']):
    for n, lab in nodes.items():
        if pat in lab:
            print('NODE', n, lab[:100].replace('\\n', ' | '))
            for p in pred.get(n, []):
                print('   <-', p, nodes.get(p, '?')[:100].replace('\\n', ' | '))
