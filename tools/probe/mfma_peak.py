"""Sustained rate of v_mfma_f32_32x32x16_bf16 with nothing else going on, as a function of how many CUs run it, the operand data
(zeros / random) and the number of independent accumulators.  Answers: how far below the 2.5 PFLOP/s data-sheet peak is the chip when
every matrix pipe is busy (power management), i.e. what roof can a bf16x3 kernel actually reach?"""
import ctypes as C, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/libmfma_peak.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', os.path.join(here, 'mfma_peak.hip'), '-o', so])
lib = C.CDLL(so)
p = lambda t: C.c_void_p(t.data_ptr())
iters = 20000
for data in ('zeros', 'random'):
    src = torch.zeros(4096, device='cuda') if data == 'zeros' else torch.randn(4096, device='cuda')
    for blocks in (64, 256, 512, 1024):
        for nacc in (1, 2, 4):
            out = torch.empty(blocks * 256, device='cuda')
            ticks = torch.zeros(blocks, dtype=torch.int64, device='cuda')
            lib.mfma_run(p(src), p(out), blocks, 200, nacc, p(ticks), None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.mfma_run(p(src), p(out), blocks, iters, nacc, p(ticks), None); e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            n_mfma = blocks * 4 * iters * nacc
            tf = n_mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12
            tk = float(ticks.double().median())
            print('%-6s blocks %4d (%.2f per CU) acc %d: %.3f ms  %7.1f TFLOP/s bf16 (= %5.1f fp32-equivalent at 6 products)  %5.1f s_memtime ticks per MFMA per wave, %.0f ticks/us'
                  % (data, blocks, blocks / 256.0, nacc, ms, tf, tf / 6, tk / (iters * nacc), tk / (ms * 1e3)), flush=True)
