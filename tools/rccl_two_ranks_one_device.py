"""Does RCCL accept two ranks on ONE device?  (a two-process RCCL test on one card, if it does.)  Starts two processes that
both use cuda:0, opens an nccl process group and tries one all-reduce; prints what happened.  python tools/rccl_two_ranks_one_device.py"""
import os, subprocess, sys, json
if "RANK" not in os.environ:
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in ps:
        try:
            o, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill(); o, _ = p.communicate(); o += "\n[timed out]"
        outs.append((p.returncode, o[-1500:]))
    print(json.dumps({"two_ranks_on_one_device": [{"rc": rc, "tail": o} for rc, o in outs]}, indent=1))
    sys.exit(0)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", os.environ["RANK"], "all_reduce over two ranks on one device WORKED:", t.tolist())
    dist.destroy_process_group()
except Exception as e:
    print("rank", os.environ["RANK"], "FAILED:", type(e).__name__, str(e)[-900:])
    sys.exit(3)
