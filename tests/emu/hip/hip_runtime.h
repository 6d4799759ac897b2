// hipemu -- TEST INFRASTRUCTURE ONLY.  A minimal CPU stand-in for <hip/hip_runtime.h> so that the product's
// pure-HIP sources (rangedet_amd/csrc/*.hip) can be compiled by a host compiler (clang++ -x c++ -Itests/emu)
// and executed on CPU *in tests only* -- indexing, LDS layouts, MFMA fragment maps, barriers and the C-ABI
// host logic get exercised (under ASan if wanted) in the CPU-only CI tier.  The product never loads the
// resulting library (rangedet_amd/lib.py only ever opens librangedet_hip.so and fails loudly without it).
//
// Execution model: every GPU thread of a block is a ucontext fiber on one OS thread, scheduled round-robin;
// __syncthreads() and the wave-level collectives (shuffles, ballot, MFMA) are rendezvous points.  Blocks of a
// grid run one after another.  wave = 64 lanes, MFMA fragment maps as documented for gfx950.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIPEMU 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

typedef int hipError_t;
typedef void* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
struct hipemuEvent { std::chrono::steady_clock::time_point t; };
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemset2DAsync(void* p, size_t pitch, int v, size_t w, size_t h, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memset((char*)p + r * pitch, v, w);
  return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }

// ---- vector types ---------------------------------------------------------------------------------
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
inline float2 make_float2(float a, float b) { return {a, b}; }
inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }
inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return {a, b, c, d}; }
inline int2 make_int2(int a, int b) { return {a, b}; }
inline int4 make_int4(int a, int b, int c, int d) { return {a, b, c, d}; }

namespace hipemu {
struct PendingDma { const void* src; void* dst; unsigned size; };
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  uint3_ tid;
  bool done = false;
  int wave = 0, lane = 0;
  std::vector<PendingDma> dmaq;   // LDS-DMA transfers issued and not yet retired (late-DMA mode only), oldest first
  size_t dma_head = 0;
};
// LDS-DMA completion model.  0 (default): a transfer lands when it is issued -- the EARLIEST the hardware allows (exposes ring slots
// overwritten while still being read).  1 ("late", hipemu_set_dma_late / HIPEMU_DMA_LATE=1): a transfer lands only when an s_waitcnt
// vmcnt(N) of its own wave forces it, i.e. when more than N younger LDS-DMA instructions of that lane's wave have been issued -- the
// LATEST the hardware allows.  A counted wait that is too weak then reads stale LDS deterministically, on the CPU.  (Only LDS-DMA
// instructions are counted: other vector-memory instructions in the hardware queue can only make a wait stricter.  The kernels issue
// their DMA instructions wave-uniformly -- padding lanes fetch from a zero page -- so the per-lane queue equals the wave's.)
inline int& dma_late() {
  static int v = [] { const char* e = getenv("HIPEMU_DMA_LATE"); return e ? atoi(e) : 0; }();
  return v;
}
inline void dma_retire(Fiber* f, size_t keep) {
  while (f->dmaq.size() - f->dma_head > keep) {
    const PendingDma& d = f->dmaq[f->dma_head++];
    memcpy(d.dst, d.src, d.size);
  }
  if (f->dma_head == f->dmaq.size()) { f->dmaq.clear(); f->dma_head = 0; }
}
struct Wave {
  int count = 0, alive = 0;
  unsigned gen = 0;
  alignas(64) unsigned char xchg[64][192];
};
struct Block {
  std::vector<Fiber> f;
  std::vector<Wave> w;
  ucontext_t sched;
  int alive = 0, bar_count = 0;
  unsigned bar_gen = 0;
  unsigned long progress = 0;
  uint3_ bid, bdim, gdim;
  std::function<void()> body;
  std::vector<unsigned char> dyn;
};
inline Block*& blk() { static thread_local Block* b = nullptr; return b; }
inline Fiber*& cur() { static thread_local Fiber* f = nullptr; return f; }
inline void yield() { swapcontext(&cur()->ctx, &blk()->sched); }
inline void trampoline() {
  Block* b = blk();
  b->body();
  Fiber* f = cur();
  dma_retire(f, 0);
  f->done = true;
  b->alive--;
  b->w[f->wave].alive--;
  b->progress++;
  // a thread that exits releases barriers the remaining threads are parked on
  if (b->alive > 0 && b->bar_count == b->alive) { b->bar_count = 0; b->bar_gen++; }
  Wave& w = b->w[f->wave];
  if (w.alive > 0 && w.count == w.alive) { w.count = 0; w.gen++; }
  swapcontext(&f->ctx, &b->sched);
}
inline void block_barrier() {
  Block* b = blk();
  unsigned gen = b->bar_gen;
  if (++b->bar_count == b->alive) { b->bar_count = 0; b->bar_gen++; b->progress++; return; }
  while (b->bar_gen == gen) yield();
}
inline void wave_barrier() {
  Block* b = blk();
  Wave& w = b->w[cur()->wave];
  unsigned gen = w.gen;
  if (++w.count == w.alive) { w.count = 0; w.gen++; b->progress++; return; }
  while (w.gen == gen) yield();
}
inline void run_block(Block& b, int nthreads) {
  blk() = &b;
  b.f.resize(nthreads);
  b.w.assign((nthreads + 63) / 64, Wave());
  b.alive = nthreads;
  for (int t = 0; t < nthreads; ++t) {
    Fiber& f = b.f[t];
    f.done = false;
    f.dmaq.clear();
    f.dma_head = 0;
    f.stack.resize(256 * 1024);
    f.tid.x = t % b.bdim.x;
    f.tid.y = (t / b.bdim.x) % b.bdim.y;
    f.tid.z = t / (b.bdim.x * b.bdim.y);
    f.wave = t / 64;
    f.lane = t % 64;
    b.w[f.wave].alive++;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
  }
  while (b.alive > 0) {
    unsigned long before = b.progress;
    for (int t = 0; t < nthreads; ++t) {
      if (b.f[t].done) continue;
      cur() = &b.f[t];
      swapcontext(&b.sched, &b.f[t].ctx);
    }
    if (b.alive > 0 && b.progress == before) {
      fprintf(stderr, "hipemu: deadlock (divergent barrier / collective) in block (%u,%u,%u)\n", b.bid.x, b.bid.y, b.bid.z);
      abort();
    }
  }
  blk() = nullptr;
  cur() = nullptr;
}
inline unsigned char* dyn_smem() { return blk()->dyn.data(); }
template <class F>
inline void launch(dim3 grid, dim3 block, size_t shmem, F&& body) {
  static thread_local Block b;
  b.body = body;
  b.dyn = std::vector<unsigned char>(shmem);  // exact size on the heap: ASan sees LDS overruns
  b.bdim = {block.x, block.y, block.z};
  b.gdim = {grid.x, grid.y, grid.z};
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        b.bid = {x, y, z};
        b.bar_count = 0;
        run_block(b, block.x * block.y * block.z);
      }
}
template <class T>
inline T exchange(T v, int src_lane) {  // every alive lane of the wave calls this
  static_assert(sizeof(T) <= 192, "exchange payload");
  Wave& w = blk()->w[cur()->wave];
  memcpy(w.xchg[cur()->lane], &v, sizeof(T));
  wave_barrier();
  T r;
  memcpy(&r, w.xchg[src_lane & 63], sizeof(T));
  wave_barrier();
  return r;
}
}  // namespace hipemu

#define threadIdx (hipemu::cur()->tid)
#define blockIdx (hipemu::blk()->bid)
#define blockDim (hipemu::blk()->bdim)
#define gridDim (hipemu::blk()->gdim)
#define warpSize 64
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::dyn_smem();
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

inline void __syncthreads() { hipemu::block_barrier(); }
template <class T> inline T __shfl(T v, int src, int width = 64) {
  int lane = hipemu::cur()->lane;
  int base = lane & ~(width - 1);
  return hipemu::exchange(v, base + (src & (width - 1)));
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) { return hipemu::exchange(v, hipemu::cur()->lane ^ m); }
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = hipemu::cur()->lane;
  int src = lane + (int)d;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return hipemu::exchange(v, src);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = hipemu::cur()->lane;
  int src = lane - (int)d;
  if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return hipemu::exchange(v, src);
}
inline unsigned long long __ballot(int pred) {
  hipemu::Wave& w = hipemu::blk()->w[hipemu::cur()->wave];
  hipemu::Block* b = hipemu::blk();
  int lane = hipemu::cur()->lane;
  int wv = hipemu::cur()->wave;
  w.xchg[lane][0] = pred ? 1 : 0;
  hipemu::wave_barrier();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) {
    int t = wv * 64 + l;
    if (t < (int)b->f.size() && !b->f[t].done && w.xchg[l][0]) m |= 1ull << l;
  }
  hipemu::wave_barrier();
  return m;
}
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline int __all(int p) { return __ballot(!p) == 0ull; }
inline int __any(int p) { return __ballot(p) != 0ull; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __builtin_amdgcn_readfirstlane(int v) { return hipemu::exchange(v, 0); }
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_barrier(); }
inline void __builtin_amdgcn_s_barrier() { hipemu::block_barrier(); }
inline void __builtin_amdgcn_fence(int, const char*, ...) {}
// s_waitcnt immediate (gfx9): vmcnt = imm[3:0] | imm[15:14] << 4.  Early mode: nothing to do.  Late mode: retire this lane's oldest
// LDS-DMA transfers until at most vmcnt are outstanding.
inline void __builtin_amdgcn_s_waitcnt(int imm) {
  if (hipemu::dma_late()) hipemu::dma_retire(hipemu::cur(), (size_t)((imm & 15) | (((imm >> 14) & 3) << 4)));
}
// LDS-DMA: destination = wave-uniform LDS base + lane * size
template <class SrcPtr, class DstPtr>
inline void __builtin_amdgcn_global_load_lds(SrcPtr src, DstPtr lds_base, unsigned size, int offset, unsigned) {
  void* dst = (unsigned char*)(uintptr_t)lds_base + offset + (size_t)hipemu::cur()->lane * size;
  if (hipemu::dma_late()) hipemu::cur()->dmaq.push_back({(const void*)(uintptr_t)src, dst, size});
  else memcpy(dst, (const void*)(uintptr_t)src, size);
}
extern "C" __attribute__((visibility("default"), used)) inline void hipemu_set_dma_late(int v) { hipemu::dma_late() = v; }
extern "C" __attribute__((visibility("default"), used)) inline int hipemu_get_dma_late() { return hipemu::dma_late(); }
inline bool isinf(float v) { return std::isinf(v); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline unsigned long long wall_clock64() { return 0; }
inline unsigned long long __builtin_readcyclecounter_emu() { return 0; }
inline void __threadfence() {}

// sequential blocks, cooperative fibers: plain read-modify-write is atomic enough
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }

inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int __mul24(int a, int b) { return (int)((unsigned)(((a << 8) >> 8)) * (unsigned)(((b << 8) >> 8))); }   // low 32 bits of the signed 24-bit product
inline float __fdividef(float a, float b) { return a / b; }
inline float __expf(float x) { return expf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }

// ---- MFMA emulation (gfx950 fragment maps) --------------------------------------------------------
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef short hipemu_s16x8 __attribute__((ext_vector_type(8)));
namespace hipemu {
inline float bf16_bits_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
struct MfmaPkt32 { float c[16]; float a[8]; float b[8]; };
// generic: A[m][k] held by lane (m + 32*(k/KL)), element k%KL ; B[k][n] by lane (n + 32*(k/KL)), element k%KL
// C/D[m][n] by lane (n + 32*((m>>2)&1)), reg (m&3) + 4*(m>>3)
template <int KL>
inline hipemu_f32x16 mfma32(const float* a, const float* b, hipemu_f32x16 c) {
  Wave& w = blk()->w[cur()->wave];
  int lane = cur()->lane;
  MfmaPkt32 pk;
  for (int i = 0; i < 16; ++i) pk.c[i] = c[i];
  for (int i = 0; i < KL; ++i) { pk.a[i] = a[i]; pk.b[i] = b[i]; }
  memcpy(w.xchg[lane], &pk, sizeof(pk));
  wave_barrier();
  hipemu_f32x16 d;
  int n = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 2 * KL; ++k) {
      const MfmaPkt32* pa = (const MfmaPkt32*)w.xchg[m + 32 * (k / KL)];
      const MfmaPkt32* pb = (const MfmaPkt32*)w.xchg[n + 32 * (k / KL)];
      acc = fmaf(pa->a[k % KL], pb->b[k % KL], acc);
    }
    d[r] = acc;
  }
  wave_barrier();
  return d;
}
}  // namespace hipemu
inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hipemu_s16x8 a, hipemu_s16x8 b, hipemu_f32x16 c, int, int, int) {
  float fa[8], fb[8];
  for (int i = 0; i < 8; ++i) { fa[i] = hipemu::bf16_bits_to_f32((unsigned short)a[i]); fb[i] = hipemu::bf16_bits_to_f32((unsigned short)b[i]); }
  return hipemu::mfma32<8>(fa, fb, c);
}
typedef _Float16 hipemu_h16x8 __attribute__((ext_vector_type(8)));
inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(hipemu_h16x8 a, hipemu_h16x8 b, hipemu_f32x16 c, int, int, int) {
  float fa[8], fb[8];
  for (int i = 0; i < 8; ++i) { fa[i] = (float)a[i]; fb[i] = (float)b[i]; }
  return hipemu::mfma32<8>(fa, fb, c);
}
// 16 x 16 x 32 (gfx950): A[m][k] lane m + 16*(k/8) element k%8; B[k][n] lane n + 16*(k/8) element k%8; D[m][n] lane n + 16*(m/4) reg m%4
// (the map tools/micro/mfma16_probe.hip confirms on the GPU)
namespace hipemu {
struct MfmaPkt16 { float a[8]; float b[8]; };
inline hipemu_f32x4 mfma16(const float* a, const float* b, hipemu_f32x4 c) {
  Wave& w = blk()->w[cur()->wave];
  int lane = cur()->lane;
  MfmaPkt16 pk;
  for (int i = 0; i < 8; ++i) { pk.a[i] = a[i]; pk.b[i] = b[i]; }
  memcpy(w.xchg[lane], &pk, sizeof(pk));
  wave_barrier();
  hipemu_f32x4 d;
  int n = lane & 15, q = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    int m = 4 * q + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k) {
      const MfmaPkt16* pa = (const MfmaPkt16*)w.xchg[m + 16 * (k / 8)];
      const MfmaPkt16* pb = (const MfmaPkt16*)w.xchg[n + 16 * (k / 8)];
      acc = fmaf(pa->a[k % 8], pb->b[k % 8], acc);
    }
    d[r] = acc;
  }
  wave_barrier();
  return d;
}
}  // namespace hipemu
inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(hipemu_s16x8 a, hipemu_s16x8 b, hipemu_f32x4 c, int, int, int) {
  float fa[8], fb[8];
  for (int i = 0; i < 8; ++i) { fa[i] = hipemu::bf16_bits_to_f32((unsigned short)a[i]); fb[i] = hipemu::bf16_bits_to_f32((unsigned short)b[i]); }
  return hipemu::mfma16(fa, fb, c);
}
inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_f16(hipemu_h16x8 a, hipemu_h16x8 b, hipemu_f32x4 c, int, int, int) {
  float fa[8], fb[8];
  for (int i = 0; i < 8; ++i) { fa[i] = (float)a[i]; fb[i] = (float)b[i]; }
  return hipemu::mfma16(fa, fb, c);
}
inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
  return hipemu::mfma32<1>(&a, &b, c);
}
