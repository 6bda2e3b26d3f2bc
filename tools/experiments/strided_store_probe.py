"""Times attic/strided_store_probe.hip on an MI355X: GB/s of a read + write stream, contiguous against 16-B slots at a 32-B stride."""
import ctypes as C
import os
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
lib = C.CDLL(os.path.join(ROOT, "tools", "exp_libs", "libprobe.so"))
lib.probe.restype = C.c_int
lib.probe.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
for blocks in (256, 1024, 4096):
    iters = 196 * 256 // blocks * 2
    per_wg = iters * 32 * 1024
    src = torch.randint(0, 255, (blocks * per_wg,), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for mode in (0, 1):
        for _ in range(2):
            assert lib.probe(src.data_ptr(), dst.data_ptr(), per_wg, iters, mode, blocks, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(src, dst)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for a, b in ev:
            a.record(); lib.probe(src.data_ptr(), dst.data_ptr(), per_wg, iters, mode, blocks, st); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)[3] * 1e-3
        print(f"blocks {blocks:5d} mode {mode}: {t * 1e6:8.1f} us  {2 * src.numel() / t / 1e12:.2f} TB/s (read + write, {src.numel() / 1e9:.2f} GB each way)", flush=True)
    del src, dst
