"""N ranks on ONE GPU (gloo): open the p2p exchange, run its self-test, print what failed.  torchrun --nproc-per-node N tools/p2p_selftest_lab.py"""
import os, sys, datetime, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NGCF_P2P_TIMEOUT_MS", "20000")
torch.cuda.set_device(0)
dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=3))
from seoul_tourism_recommendation_ngcf_amd import dist as nd
ex = nd.P2PExchange(None, torch.device("cuda", 0), int(os.environ.get("LAB_FLOATS", 1 << 20)))
ok = ex.selftest()
print(f"rank {dist.get_rank()}: selftest {ok} {getattr(ex, 'last_error', None)}", flush=True)
dist.barrier()
ex.close()
dist.destroy_process_group()
