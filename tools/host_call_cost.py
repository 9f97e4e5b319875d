"""Host-side cost per call of the ops layer and of its ingredients (microseconds, tiny tensors: the GPU side is negligible):
   python tools/host_call_cost.py   - MI355X box, round 5: ops.dwconv3x3 10.8, raw ctypes call 4.0, torch add 4.3, F.conv2d (MIOpen) 46."""
import time, torch, sys
sys.path.insert(0, '/root/repo')
import wave_mamba_amd as wm
dev = torch.device('cuda', 0)
x = torch.randn(1, 8, 8, 8, device=dev); w = torch.randn(8, 1, 3, 3, device=dev)
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n; torch.cuda.synchronize(); return dt * 1e6
def g():
    with torch.cuda.device(dev): pass
print("with torch.cuda.device: %.2f us" % t(g))
print("torch.cuda.current_device(): %.2f us" % t(lambda: torch.cuda.current_device()))
print("torch.cuda.current_stream().cuda_stream: %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("torch.empty_like: %.2f us" % t(lambda: torch.empty_like(x)))
print("x.contiguous().float(): %.2f us" % t(lambda: x.contiguous().float()))
print("ops.dwconv3x3 (tiny): %.2f us" % t(lambda: wm.ops.dwconv3x3(x, w, None, 'none')))
print("torch add (tiny): %.2f us" % t(lambda: x + x))
print("F.conv2d tiny: %.2f us" % t(lambda: torch.nn.functional.conv2d(x, w, None, padding=1, groups=8)))
lib = wm._lib.load()
y = torch.empty_like(x)
print("raw ctypes wm_dwconv3x3_fwd: %.2f us" % t(lambda: lib.wm_dwconv3x3_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 1, 8, 8, 8, 0, 0, 0)))
