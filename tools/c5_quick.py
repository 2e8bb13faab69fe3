#!/usr/bin/env python
"""c5 object of bench.py alone:  python tools/c5_quick.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
class A: pass
a = A(); a.precision = "bf16"; a.index_rows = 10000; a.k = 10
print(json.dumps(bench.c5_extras(a, torch.device("cuda:0")), indent=1))
