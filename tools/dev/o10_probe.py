import os, sys, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, "/root/repo")
import torch, bench
from deepfilternet_amd.libdf import DF
dev = torch.device("cuda", 0)
st = DF(48000, 960, 480, 32, 2)
print(json.dumps(bench.bench_df_apply_o10(dev, st, 256, 1002)))
