"""LDS-DMA throughput per CU (sm_peak_lds_dma): bytes per clock and CU for 1 / 2 / 4 / 8 / 16 issuing waves with 8 / 16 / 32 loads in flight each, from an L2-resident
window (64 KiB per workgroup) and from HBM (4 MiB per workgroup, 1 GiB in total), one workgroup per CU."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "opensearch-sparse-model-tuning-sample_amd"))
from sparse_hip import lib as L
src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda").random_(0, 255)
sink = torch.zeros(1024, device="cuda")
clk_mhz = float(os.environ.get("CLK_MHZ", "2400"))
blocks = 256
for span, what in ((64 << 10, "L2-resident"), (4 << 20, "HBM / Infinity Cache")):
    for waves in (1, 2, 4, 8, 16):
        row = []
        for depth in (8, 16) if waves == 16 else (8, 16, 32):
            iters = 4096 // waves * 4
            st = L.stream_ptr()
            L.call("sm_peak_lds_dma", L.ptr(src), span, blocks, waves, depth, 64, L.ptr(sink), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.call("sm_peak_lds_dma", L.ptr(src), span, blocks, waves, depth, iters, L.ptr(sink), st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            nbytes = blocks * waves * iters * 1024
            row.append(f"depth {depth}: {nbytes / ms / 1e9:6.2f} TB/s = {nbytes / blocks / (ms * 1e-3 * clk_mhz * 1e6):5.1f} B/clk/CU")
        print(f"{what:22s} waves {waves}: " + " | ".join(row))
