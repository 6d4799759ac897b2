"""Which instruction class of a co-resident wave makes v_pk_mul_f32 with swapped src1 halves go wrong (tools/micro/victim.hip)."""
import ctypes
import os
import sys

import torch

V = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvictim.so"))
V.pkform_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
V.aggr_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
src = torch.randn(64 * 262144 + 512 * 256 * 4 + 1024, device="cuda")
sink = torch.zeros(512 * 256, device="cuda")
cnt = torch.zeros(80, device="cuda", dtype=torch.int32)
kinds = ["v_mfma_f32_32x32x16_bf16", "ds_read_b128", "global_load_lds_dwordx4", "global_load_dwordx4 (nt)", "v_pk_max_i16 + v_cvt_pk_bf16_f32", "s_barrier",
         "v_fma_f32", "ds_write_b128", "v_mfma_f32_16x16x32_bf16", "v_pk_mul_f32 (default form)"]
iters = {0: 40000, 1: 60000, 2: 8000, 3: 8000, 4: 200000, 5: 100000, 6: 200000, 7: 60000, 8: 80000, 9: 200000}
for kind, name in enumerate(kinds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    V.aggr_run(kind, src.data_ptr(), sink.data_ptr(), 512, iters[kind], torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    cnt.zero_()
    torch.cuda.synchronize()
    for r in range(reps):
        V.aggr_run(kind, src.data_ptr(), sink.data_ptr(), 512, iters[kind], s1.cuda_stream)
        for _ in range(3):
            V.pkform_run(0, cnt.data_ptr(), 1024, 1000, s2.cuda_stream)
        torch.cuda.synchronize()
    c = cnt.cpu().numpy().reshape(10, 2, 4)[0]
    print("co-resident %-34s (%6.0f us alone): victim mismatches by lane quarter  lo %s | hi %s" % (name, us, c[0].tolist(), c[1].tolist()), flush=True)

# ---- 16-bit packed forms with the same crossing, next to the MFMA spin kernel --------------------------------------------------
V.pk16_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n16 = ["v_pk_mul_f16 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_fma_f16 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_pk_max_i16 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f16 (default)"]
cnt.zero_()
torch.cuda.synchronize()
for r in range(reps):
    V.aggr_run(8, src.data_ptr(), sink.data_ptr(), 512, iters[8], s1.cuda_stream)
    for pat in range(4):
        V.pk16_run(pat, cnt.data_ptr(), 1024, 1000, s2.cuda_stream)
    V.pkform_run(0, cnt.data_ptr() + 4 * 32, 1024, 1000, s2.cuda_stream)        # the fp32 form in the same overlap, as the positive control
    torch.cuda.synchronize()
c = cnt.cpu().numpy().reshape(10, 2, 4)
for p in range(4):
    print("next to v_mfma_f32_16x16x32_bf16: %-46s lo %s | hi %s" % (n16[p], c[p, 0].tolist(), c[p, 1].tolist()))
print("next to v_mfma_f32_16x16x32_bf16: %-46s lo %s | hi %s" % ("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (control)", c[4, 0].tolist(), c[4, 1].tolist()), flush=True)
