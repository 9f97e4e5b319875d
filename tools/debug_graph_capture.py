"""Which part of the profiling hooks invalidates a later stream capture?  STAGE = none | enable | step | collect | torch_events"""
import os, sys, torch, traceback
sys.path.insert(0, '/root/repo')
import wave_mamba_amd as wm, bench
dev = torch.device('cuda', 0)
net = bench.build_model(dev)
H, W = (int(v) for v in os.environ.get("HW", "512x768").split("x"))
x = bench.pad_to(torch.rand(1, 3, H, W)).to(dev)
unet = net.restoration_network
unet.two_streams = os.environ.get("TWO", "0") == "1"
def step():
    with torch.no_grad():
        return unet(x)
stage = os.environ.get("STAGE", "none")
if os.environ.get("PRETWO") == "1":
    two = unet.two_streams; unet.two_streams = True; step(); torch.cuda.synchronize(); unet.two_streams = two
step(); torch.cuda.synchronize()
if stage in ("enable", "step", "collect"):
    wm.ops.prof_enable(bench.CORE_CLASSES)
    if stage in ("step", "collect"):
        step(); torch.cuda.synchronize()
    if stage == "collect":
        print({k: v for k, v in wm.ops.prof_collect().items() if v[0]})
    wm.ops.prof_enable(False)
if stage == "torch_events":
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); step(); e1.record(); torch.cuda.synchronize(); print("torch events ms", e0.elapsed_time(e1))
step(); torch.cuda.synchronize()
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    g.replay(); torch.cuda.synchronize()
    print('stage', stage, 'two', unet.two_streams, 'capture ok')
except Exception as e:
    print('stage', stage, 'two', unet.two_streams, 'FAILED', type(e).__name__, str(e)[:160])
