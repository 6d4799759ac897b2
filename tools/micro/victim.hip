// Probe kernels of the round-5 co-residency study (profiles/r05h_packed_swap_fault.txt):
//   victim_kernel   do long-lived registers / scalar, packed, division chains of a plain kernel survive next to the conv kernels?  (yes)
//   war_kernel      write-after-read probes around 64-bit VALU operands                                                            (clean)
//   pk_kernel       the instruction sequence of wnms_prep_kernel's area computation                                               (reproduces)
//   pkform_kernel   one packed-fp32 instruction per pattern: WHICH modifier combination is wrong   (second source's halves swapped)
//   aggr_kernel     one instruction class per co-resident wave: WHAT has to run on the same SIMD                          (any MFMA)
// build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/victim.hip -o tools/micro/libvictim.so
// run (through gpurun):  bash tools/micro/fault_study.sh
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));

// mode 0: scalar fma chain;  1: packed-fp32 chain;  2: division / sqrt chain;  3: divergent loop (lane-dependent trip counts)
// every thread keeps 8 values live across the whole spin and adds them to its result at the end
template <int MODE>
__global__ __launch_bounds__(256) void victim_kernel(float* __restrict__ out, int iters, float seed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float keep[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) keep[k] = seed + (float)(i * 8 + k) * 0.001f;
#pragma unroll
  for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(keep[k]));
  float a = 1.0f + (float)(i & 1023) * 1e-3f, b = 0.5f;
  f2 pa = {a, a + 0.25f}, pb = {b, b + 0.125f};
  int n = iters;
  if (MODE == 3) n = iters / 4 + ((i * 2654435761u) >> 26) * (iters / 64);
  for (int t = 0; t < n; ++t) {
    if (MODE == 0 || MODE == 3) { a = __builtin_fmaf(a, 0.9999f, b); b = __builtin_fmaf(b, 0.5f, 0.25f); }
    if (MODE == 1) { pa = pa * (f2){0.9999f, 0.9998f} + pb; pb = pb * (f2){0.5f, 0.5f} + (f2){0.25f, 0.125f}; }
    if (MODE == 2) { a = a / (b + 1.5f) + 1.0f; b = __builtin_sqrtf(a + b); }
  }
  float r = a + b + pa.x + pa.y + pb.x + pb.y;
#pragma unroll
  for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(keep[k]));
#pragma unroll
  for (int k = 0; k < 8; ++k) r += keep[k];
  out[i] = r;
}
extern "C" int victim_run(int mode, float* out, int nblocks, int iters, float seed, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(victim_kernel<0>, dim3(nblocks), dim3(256), 0, st, out, iters, seed); break;
    case 1: hipLaunchKernelGGL(victim_kernel<1>, dim3(nblocks), dim3(256), 0, st, out, iters, seed); break;
    case 2: hipLaunchKernelGGL(victim_kernel<2>, dim3(nblocks), dim3(256), 0, st, out, iters, seed); break;
    default: hipLaunchKernelGGL(victim_kernel<3>, dim3(nblocks), dim3(256), 0, st, out, iters, seed); break;
  }
  return (int)hipGetLastError();
}

// ---- write-after-read probes: a packed / 64-bit VALU op reads v21, the NEXT instruction overwrites v21 ---------------------------
// out[pattern][lane quarter] += mismatches.  PAT 0: v_pk_mov_b32, 1: v_pk_mul_f32, 2: v_mul_f32 (control), 3: v_pk_add_f32, 4: v_mov_b64
template <int PAT>
__global__ __launch_bounds__(256) void war_kernel(unsigned* __restrict__ out, int iters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned bad = 0;
  for (int t = 0; t < iters; ++t) {
    float a = 1.0f + (float)((i + t) & 255), b = 3.0f + (float)((i * 7 + t) & 255), c = 2.0f, d = 700.0f + (float)(t & 63);
    float res, expect;
    if (PAT == 0) {
      asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v24, %3\n\tv_mov_b32 v25, %4\n\ts_nop 4\n\t"
                   "v_pk_mov_b32 v[22:23], v[20:21], v[24:25] op_sel:[1,0]\n\tv_mov_b32 v21, v25\n\ts_nop 4\n\tv_mov_b32 %0, v22"
                   : "=v"(res) : "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "v24", "v25");
      expect = b;
    } else if (PAT == 1) {
      asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v24, %3\n\tv_mov_b32 v25, %4\n\ts_nop 4\n\t"
                   "v_pk_mul_f32 v[22:23], v[20:21], v[24:25]\n\tv_mov_b32 v21, v25\n\ts_nop 4\n\tv_mov_b32 %0, v23"
                   : "=v"(res) : "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "v24", "v25");
      expect = b * d;
    } else if (PAT == 2) {
      asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v24, %3\n\tv_mov_b32 v25, %4\n\ts_nop 4\n\t"
                   "v_mul_f32 v23, v21, v25\n\tv_mov_b32 v21, v25\n\ts_nop 4\n\tv_mov_b32 %0, v23"
                   : "=v"(res) : "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "v24", "v25");
      expect = b * d;
    } else if (PAT == 3) {
      asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v24, %3\n\tv_mov_b32 v25, %4\n\ts_nop 4\n\t"
                   "v_pk_add_f32 v[22:23], v[20:21], v[24:25]\n\tv_mov_b32 v21, v25\n\ts_nop 4\n\tv_mov_b32 %0, v23"
                   : "=v"(res) : "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "v24", "v25");
      expect = b + d;
    } else {
      asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v24, %3\n\tv_mov_b32 v25, %4\n\ts_nop 4\n\t"
                   "v_mov_b64 v[22:23], v[20:21]\n\tv_mov_b32 v21, v25\n\ts_nop 4\n\tv_mov_b32 %0, v23"
                   : "=v"(res) : "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "v24", "v25");
      expect = b;
    }
    bad += res != expect;
  }
  if (bad) atomicAdd(&out[PAT * 4 + ((threadIdx.x & 63) >> 4)], bad);
}
extern "C" int war_run(int pat, unsigned* out, int nblocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (pat) {
    case 0: hipLaunchKernelGGL(war_kernel<0>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    case 1: hipLaunchKernelGGL(war_kernel<1>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    case 2: hipLaunchKernelGGL(war_kernel<2>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    case 3: hipLaunchKernelGGL(war_kernel<3>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    default: hipLaunchKernelGGL(war_kernel<4>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
  }
  return (int)hipGetLastError();
}

// ---- the sequence found in wnms_prep_kernel: a 32-bit write into ONE half of a register pair, then a packed-fp32 op reading the pair with
// op_sel.  out[pat*8 + quarter] += lo-half mismatches, out[pat*8 + 4 + quarter] += hi-half mismatches.
// PAT 0: mov hi; pk_mul(filler); pk_mul op_sel   1: mov hi; pk_mul op_sel (no filler)   2: mov hi; s_nop 1; pk_mul op_sel
// PAT 3: as 0 without op_sel   4: as 0 but the pair written whole by v_mov_b64   5: mov LO half; filler; pk_mul op_sel
#define PK_SEQ(BODY)                                                                                                            \
  asm volatile("v_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\tv_mov_b32 v18, %4\n\tv_mov_b32 v19, %5\n\tv_mov_b32 v25, %6\n\t"      \
               "v_mov_b32 v26, %6\n\tv_mov_b32 v27, %2\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %5\n\ts_nop 4\n\t" BODY           \
               "\n\ts_nop 4\n\tv_mov_b32 %0, v28\n\tv_mov_b32 %1, v29"                                                          \
               : "=v"(lo), "=v"(hi) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e)                                                    \
               : "v18", "v19", "v20", "v21", "v22", "v23", "v25", "v26", "v27", "v28", "v29")
template <int PAT>
__global__ __launch_bounds__(256) void pk_kernel(unsigned* __restrict__ out, int iters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned badlo = 0, badhi = 0;
  for (int t = 0; t < iters; ++t) {
    float a = 1.0f + (float)((i + t) & 255), b = 3.0f + (float)((i * 7 + t) & 255), c = 2.0f + (float)(t & 7), d = 5.0f + (float)(i & 15),
          e = 700.0f + (float)(t & 63);
    float lo, hi, xlo, xhi;
    if (PAT == 0) {
      PK_SEQ("v_mov_b32 v21, v25\n\tv_pk_mul_f32 v[22:23], v[22:23], v[18:19]\n\t"
             "v_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]");
      xlo = -(a * d); xhi = -(e * c);
    } else if (PAT == 1) {
      PK_SEQ("v_mov_b32 v21, v25\n\tv_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]");
      xlo = -(a * d); xhi = -(e * c);
    } else if (PAT == 2) {
      PK_SEQ("v_mov_b32 v21, v25\n\ts_nop 1\n\tv_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]");
      xlo = -(a * d); xhi = -(e * c);
    } else if (PAT == 3) {
      PK_SEQ("v_mov_b32 v21, v25\n\tv_pk_mul_f32 v[22:23], v[22:23], v[18:19]\n\tv_pk_mul_f32 v[28:29], v[20:21], v[18:19]");
      xlo = a * c; xhi = e * d;
    } else if (PAT == 4) {
      PK_SEQ("v_mov_b64 v[20:21], v[26:27]\n\tv_pk_mul_f32 v[22:23], v[22:23], v[18:19]\n\t"
             "v_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]");
      xlo = -(e * d); xhi = -(a * c);
    } else {
      PK_SEQ("v_mov_b32 v20, v25\n\tv_pk_mul_f32 v[22:23], v[22:23], v[18:19]\n\t"
             "v_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]");
      xlo = -(e * d); xhi = -(b * c);
    }
    badlo += lo != xlo;
    badhi += hi != xhi;
  }
  const int qd = (threadIdx.x & 63) >> 4;
  if (badlo) atomicAdd(&out[PAT * 8 + qd], badlo);
  if (badhi) atomicAdd(&out[PAT * 8 + 4 + qd], badhi);
}
extern "C" int pk_run(int pat, unsigned* out, int nblocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (pat) {
    case 0: hipLaunchKernelGGL(pk_kernel<0>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    case 1: hipLaunchKernelGGL(pk_kernel<1>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    case 2: hipLaunchKernelGGL(pk_kernel<2>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    case 3: hipLaunchKernelGGL(pk_kernel<3>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    case 4: hipLaunchKernelGGL(pk_kernel<4>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
    default: hipLaunchKernelGGL(pk_kernel<5>, dim3(nblocks), dim3(256), 0, st, out, iters); break;
  }
  return (int)hipGetLastError();
}

// ---- which packed forms are affected (one instruction per pattern; operands set up 5+ wait states earlier) ----------------------------
#define PK1(INSTR) PK_SEQ(INSTR)
template <int PAT>
__global__ __launch_bounds__(256) void pkform_kernel(unsigned* __restrict__ out, int iters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned badlo = 0, badhi = 0;
  for (int t = 0; t < iters; ++t) {
    // v20 = a, v21 = b, v18 = c, v19 = d, v22 = c, v23 = d
    float a = 1.0f + (float)((i + t) & 255), b = 3.0f + (float)((i * 7 + t) & 255), c = 2.0f + (float)(t & 7), d = 5.0f + (float)(i & 15),
          e = 700.0f + (float)(t & 63);
    float lo, hi, xlo, xhi;
    if (PAT == 0) { PK1("v_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]"); xlo = a * d; xhi = b * c; }
    else if (PAT == 1) { PK1("v_pk_mul_f32 v[28:29], v[20:21], v[18:19] neg_lo:[0,1] neg_hi:[0,1]"); xlo = -(a * c); xhi = -(b * d); }
    else if (PAT == 2) { PK1("v_pk_add_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]"); xlo = a + d; xhi = b + c; }
    else if (PAT == 3) { PK1("v_pk_fma_f32 v[28:29], v[20:21], v[18:19], v[22:23] op_sel:[0,1,0] op_sel_hi:[1,0,1]"); xlo = __builtin_fmaf(a, d, c); xhi = __builtin_fmaf(b, c, d); }
    else if (PAT == 4) { PK1("v_pk_mov_b32 v[28:29], v[20:21], v[18:19] op_sel:[1,0]"); xlo = b; xhi = c; }
    else if (PAT == 5) { PK1("v_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[1,0] op_sel_hi:[0,1]"); xlo = b * c; xhi = a * d; }
    else if (PAT == 6) { PK1("v_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[1,1] op_sel_hi:[1,1]"); xlo = b * d; xhi = b * d; }
    else if (PAT == 7) { PK1("v_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,0] op_sel_hi:[0,0]"); xlo = a * c; xhi = a * c; }
    else if (PAT == 8) { PK1("v_pk_mul_f32 v[28:29], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,1]"); xlo = a * d; xhi = b * d; }
    else { PK1("v_pk_mul_f32 v[28:29], v[20:21], v[18:19]"); xlo = a * c; xhi = b * d; }
    (void)e;
    badlo += lo != xlo;
    badhi += hi != xhi;
  }
  const int qd = (threadIdx.x & 63) >> 4;
  if (badlo) atomicAdd(&out[PAT * 8 + qd], badlo);
  if (badhi) atomicAdd(&out[PAT * 8 + 4 + qd], badhi);
}
template <int P>
static void pkform_go(unsigned* out, int nb, int it, hipStream_t st) { hipLaunchKernelGGL(pkform_kernel<P>, dim3(nb), dim3(256), 0, st, out, it); }
extern "C" int pkform_run(int pat, unsigned* out, int nblocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (pat) {
    case 0: pkform_go<0>(out, nblocks, iters, st); break; case 1: pkform_go<1>(out, nblocks, iters, st); break;
    case 2: pkform_go<2>(out, nblocks, iters, st); break; case 3: pkform_go<3>(out, nblocks, iters, st); break;
    case 4: pkform_go<4>(out, nblocks, iters, st); break; case 5: pkform_go<5>(out, nblocks, iters, st); break;
    case 6: pkform_go<6>(out, nblocks, iters, st); break; case 7: pkform_go<7>(out, nblocks, iters, st); break;
    case 8: pkform_go<8>(out, nblocks, iters, st); break; default: pkform_go<9>(out, nblocks, iters, st); break;
  }
  return (int)hipGetLastError();
}

// ---- which instruction class of a co-resident wave does it (aggr_kernel<KIND> spins on ONE class; victim: pkform_kernel<0>) ------------
typedef short s16x8v __attribute__((ext_vector_type(8)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// KIND 0: v_mfma_f32_32x32x16_bf16   1: ds_read_b128   2: global_load_lds_dwordx4   3: global_load_dwordx4   4: v_pk_max_i16 + v_cvt_pk_bf16_f32
//      5: s_barrier   6: v_fma_f32   7: ds_write_b128   8: v_mfma_f32_16x16x32_bf16   9: v_pk_mul_f32 (default form)
template <int KIND>
__global__ __launch_bounds__(256, 2) void aggr_kernel(const float* __restrict__ src, float* __restrict__ sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x;
  for (int k = tid; k < 8192; k += 256) lds[k] = (float)k;
  __syncthreads();
  f32x16v acc = {};
  f32x4v acc4 = {};
  s16x8v va, vb;
  for (int k = 0; k < 8; ++k) { va[k] = (short)(0x3f80 + tid + k); vb[k] = (short)(0x3f00 + k); }
  float x = 1.0f + tid, y = 0.5f;
  f32x4v ld4 = {};
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v p = {x, y}, q = {0.999f, 1.001f};
  unsigned u = 0x3f803f80u + tid, w = 0x3f003f00u;
  const float* gp = src + (size_t)(blockIdx.x * 256 + tid) * 4;
  for (int t = 0; t < iters; ++t) {
    if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
    if (KIND == 8) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc4, 0, 0, 0);
    if (KIND == 1) { f32x4v v = *(volatile f32x4v*)&lds[((tid + t) & 2047) * 4]; ld4 += v; }
    if (KIND == 2) {
      const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds) + ((t & 3) << 12);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(4)" :: "v"(gp), "s"(la) : "memory");
    }
    if (KIND == 3) { f32x4v v = __builtin_nontemporal_load((const f32x4v*)(gp + (size_t)(t & 63) * 262144)); ld4 += v; }
    if (KIND == 4) { asm volatile("v_pk_max_i16 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %1, %2, %3" : "+v"(u), "+v"(w) : "v"(x), "v"(y)); }
    if (KIND == 5) __syncthreads();
    if (KIND == 6) { x = __builtin_fmaf(x, 0.9999f, y); y = __builtin_fmaf(y, 0.5f, 0.25f); }
    if (KIND == 7) { *(volatile f32x4v*)&lds[((tid + t) & 2047) * 4] = ld4; }
    if (KIND == 9) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q)); }
  }
  if (KIND == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float r = x + y + ld4[0] + ld4[1] + ld4[2] + ld4[3] + acc4[0] + acc4[3] + p[0] + p[1] + __uint_as_float(u) + __uint_as_float(w);
  for (int k = 0; k < 16; ++k) r += acc[k];
  if (r == 12345.678f) sink[blockIdx.x * 256 + tid] = r + lds[tid];
}
template <int K>
static void aggr_go(const float* src, float* sink, int nb, int it, hipStream_t st) { hipLaunchKernelGGL(aggr_kernel<K>, dim3(nb), dim3(256), 0, st, src, sink, it); }
extern "C" int aggr_run(int kind, const float* src, float* sink, int nblocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case 0: aggr_go<0>(src, sink, nblocks, iters, st); break; case 1: aggr_go<1>(src, sink, nblocks, iters, st); break;
    case 2: aggr_go<2>(src, sink, nblocks, iters, st); break; case 3: aggr_go<3>(src, sink, nblocks, iters, st); break;
    case 4: aggr_go<4>(src, sink, nblocks, iters, st); break; case 5: aggr_go<5>(src, sink, nblocks, iters, st); break;
    case 6: aggr_go<6>(src, sink, nblocks, iters, st); break; case 7: aggr_go<7>(src, sink, nblocks, iters, st); break;
    case 8: aggr_go<8>(src, sink, nblocks, iters, st); break; default: aggr_go<9>(src, sink, nblocks, iters, st); break;
  }
  return (int)hipGetLastError();
}

// ---- the 16-bit packed forms with the same operand crossing (documentation of the lint's scope: are they affected too?) ---------------
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef short s2v __attribute__((ext_vector_type(2)));
// PAT 0: v_pk_mul_f16 swapped src1   1: v_pk_fma_f16 swapped src1   2: v_pk_max_i16 swapped src1   3: v_pk_mul_f16 default (control)
template <int PAT>
__global__ __launch_bounds__(256) void pk16_kernel(unsigned* __restrict__ out, int iters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned badlo = 0, badhi = 0;
  for (int t = 0; t < iters; ++t) {
    h2v a = {(_Float16)(1 + ((i + t) & 7)), (_Float16)(2 + ((i * 3 + t) & 7))}, b = {(_Float16)(3 + (t & 3)), (_Float16)(5 + (i & 3))}, c = {(_Float16)1, (_Float16)2};
    h2v r, x;
    if (PAT == 0) { asm volatile("s_nop 4\n\tv_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 4" : "=v"(r) : "v"(a), "v"(b)); x = (h2v){a.x * b.y, a.y * b.x}; }
    else if (PAT == 1) { asm volatile("s_nop 4\n\tv_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 4" : "=v"(r) : "v"(a), "v"(b), "v"(c)); x = (h2v){a.x * b.y + c.x, a.y * b.x + c.y}; }
    else if (PAT == 2) {
      s2v ai = {(short)(i + t), (short)(7 * i - t)}, bi = {(short)(3 * t - i), (short)(i ^ t)}, ri;
      asm volatile("s_nop 4\n\tv_pk_max_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 4" : "=v"(ri) : "v"(ai), "v"(bi));
      badlo += ri.x != (ai.x > bi.y ? ai.x : bi.y); badhi += ri.y != (ai.y > bi.x ? ai.y : bi.x);
      continue;
    } else { asm volatile("s_nop 4\n\tv_pk_mul_f16 %0, %1, %2\n\ts_nop 4" : "=v"(r) : "v"(a), "v"(b)); x = (h2v){a.x * b.x, a.y * b.y}; }
    badlo += r.x != x.x;
    badhi += r.y != x.y;
  }
  const int qd = (threadIdx.x & 63) >> 4;
  if (badlo) atomicAdd(&out[PAT * 8 + qd], badlo);
  if (badhi) atomicAdd(&out[PAT * 8 + 4 + qd], badhi);
}
template <int P>
static void pk16_go(unsigned* out, int nb, int it, hipStream_t st) { hipLaunchKernelGGL(pk16_kernel<P>, dim3(nb), dim3(256), 0, st, out, it); }
extern "C" int pk16_run(int pat, unsigned* out, int nblocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (pat) {
    case 0: pk16_go<0>(out, nblocks, iters, st); break; case 1: pk16_go<1>(out, nblocks, iters, st); break;
    case 2: pk16_go<2>(out, nblocks, iters, st); break; default: pk16_go<3>(out, nblocks, iters, st); break;
  }
  return (int)hipGetLastError();
}

// ---- stand-alone reproducer (no Python, no torch, nothing of the library) -----------------------------------------------------------
//   hipcc --offload-arch=gfx950 -O3 -DFAULT_REPRO_MAIN tools/micro/victim.hip -o tools/micro/fault_repro && tools/micro/fault_repro
// Two streams: an MFMA spin kernel on one, single-instruction victims on the other.  Prints mismatches by lane quarter for the swapped
// second-source form and for the default form, next to the MFMA kernel and on an idle GPU.
#ifdef FAULT_REPRO_MAIN
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  float *src, *sink; unsigned* cnt;
  CK(hipMalloc(&src, (size_t)(64 * 262144 + 512 * 256 * 4 + 1024) * 4)); CK(hipMalloc(&sink, 512 * 256 * 4)); CK(hipMalloc(&cnt, 80 * 4));
  CK(hipMemset(src, 0, (size_t)(64 * 262144 + 512 * 256 * 4 + 1024) * 4));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("%s (%s), %d CUs\n", pr.name, pr.gcnArchName, pr.multiProcessorCount);
  const char* what[2] = {"next to v_mfma_f32_16x16x32_bf16 waves", "idle GPU"};
  for (int idle = 0; idle < 2; ++idle) {
    CK(hipMemset(cnt, 0, 80 * 4));
    for (int r = 0; r < 5; ++r) {
      if (!idle) aggr_run(8, src, sink, 512, 80000, s1);
      for (int k = 0; k < 3; ++k) { pkform_run(0, cnt, 1024, 1000, s2); pkform_run(9, cnt, 1024, 1000, s2); }
      CK(hipDeviceSynchronize());
    }
    unsigned h[80]; CK(hipMemcpy(h, cnt, sizeof h, hipMemcpyDeviceToHost));
    printf("%-40s v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]: lo [%u, %u, %u, %u] hi [%u, %u, %u, %u]   default form: lo [%u, %u, %u, %u] hi [%u, %u, %u, %u]"
           "   (mismatches of %.1e executions per lane quarter)\n", what[idle], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7],
           h[72], h[73], h[74], h[75], h[76], h[77], h[78], h[79], 5.0 * 3 * 1024 * 64 * 1000);
  }
  return 0;
}
#endif
