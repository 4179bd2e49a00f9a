# the real LSTM predictor alone (bench.lstm_cell_extra): python profiles/tools/lstm_bench.py
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
print("lstm", json.dumps(bench.lstm_cell_extra(torch, kv.lib)))
