// Dev micro-benchmark (round 6): what the matrix cores sustain on THIS part under the power cap, as a function of the operand
// statistics, the operand roles, the instruction shape and the operand source -- the "practical MFMA ceiling" the persistent conv
// kernels are priced against.  No library code; 512 workgroups x 4 waves (two per CU, two waves per SIMD), every
// wave runs the conv kernel's register blocking: 2 pixel fragments x 4 channel fragments = 8 independent accumulators per k-step.
//   SHAPE 0: v_mfma_f32_32x32x16_bf16 (what the conv kernels issue)      1: v_mfma_f32_16x16x32_bf16 (same FLOPs per operand byte pair)
//   ROLE  0: A = weights, B = activations (the conv kernels)              1: A = activations, B = weights
//   SRC   0: operands stay in registers      1: operands re-read from LDS every k-step (6 ds_read_b128 per 8 MFMAs = the cout-128 form)
// Data: weights N(0, 0.05) bf16; activations all zero / relu(N(0,1)) (half zeros, like real activations) / N(0,1).
// Accumulators restart from zero every 72 k-steps (one 128-channel tile).  While the kernels run a host thread samples the hwmon
// power / shader-clock files of every GPU it can read.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_power.bin tools/micro/mfma_power.hip -lpthread && tools/micro/mfma_power.bin
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <glob.h>
#include <unistd.h>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NWF = 32, NAF = 24;   // fragments (1 KB each) in the weight / activation pools

template <int SHAPE> struct Acc { typedef f32x16 T; };
template <> struct Acc<1> { typedef f32x4 T; };
template <int SHAPE> __device__ __forceinline__ typename Acc<SHAPE>::T mm(s16x8 a, s16x8 b, typename Acc<SHAPE>::T c) {
  if constexpr (SHAPE == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int SHAPE, int ROLE, int SRC, int PB = 2, int CB = 4, int ORDER = 0>   // ORDER 0: consecutive MFMAs share the weight fragment, 1: the pixel fragment
__global__ __launch_bounds__(256, 2) void k(const s16x8* __restrict__ wts, const s16x8* __restrict__ acts, float* out, long long* stamp, int blocks) {
  typedef typename Acc<SHAPE>::T acc_t;
  __shared__ s16x8 lw[NWF * 64], la[NAF * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NWF * 64; i += 256) lw[i] = wts[i];
  for (int i = tid; i < NAF * 64; i += 256) la[i] = acts[i];
  __syncthreads();
  s16x8 rw[2][CB], ra[2][PB];
  for (int s = 0; s < 2; ++s) {
    for (int j = 0; j < CB; ++j) rw[s][j] = lw[((wave * 8 + s * CB + j) % NWF) * 64 + lane];
    for (int i = 0; i < PB; ++i) ra[s][i] = la[((wave * 4 + s * PB + i) % NAF) * 64 + lane];
  }
  acc_t acc[PB][CB];
  float sink = 0.f;
  const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  int pw = wave * 8, pa = wave * 4;   // pool cursors (SRC 1)
  // k-steps between accumulator restarts: one 128-channel tile = 72 k-steps of 16 channels (32x32x16) or 36 of 32 (16x16x32)
  constexpr int KSTEPS = SHAPE == 0 ? 72 : 36 * (32 / (PB * CB));
  for (int b = 0; b < blocks; ++b) {
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if (b) sink += acc[i][j][0];
        for (int r = 0; r < (SHAPE == 0 ? 16 : 4); ++r) acc[i][j][r] = 0.f;
      }
#pragma unroll 2
    for (int s = 0; s < KSTEPS; ++s) {
      s16x8 w4[CB], a2[PB];
      if constexpr (SRC == 1) {
#pragma unroll
        for (int j = 0; j < CB; ++j) w4[j] = lw[((pw + j) & (NWF - 1)) * 64 + lane];
#pragma unroll
        for (int i = 0; i < PB; ++i) a2[i] = la[(pa + i) * 64 + lane];
        pw = (pw + CB) & (NWF - 1);
        pa = pa + PB >= NAF ? 0 : pa + PB;
      } else {
#pragma unroll
        for (int j = 0; j < CB; ++j) w4[j] = rw[s & 1][j];
#pragma unroll
        for (int i = 0; i < PB; ++i) a2[i] = ra[s & 1][i];
      }
#pragma unroll
      for (int n = 0; n < PB * CB; ++n) {
        const int j = ORDER == 0 ? n / PB : n % CB, i = ORDER == 0 ? n % PB : n / CB;
        acc[i][j] = ROLE == 0 ? mm<SHAPE>(w4[j], a2[i], acc[i][j]) : mm<SHAPE>(a2[i], w4[j], acc[i][j]);
      }
    }
  }
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  for (int i = 0; i < PB; ++i) for (int j = 0; j < CB; ++j) for (int r = 0; r < (SHAPE == 0 ? 16 : 4); ++r) sink += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = sink;
  if (tid == 0) { stamp[blockIdx.x * 2] = c1 - c0; stamp[blockIdx.x * 2 + 1] = w1 - w0; }
}

static unsigned short bf16(float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static double urand() { static unsigned long long s = 88172645463325252ull; s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }
static float nrand() { return (float)(sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand())); }

struct Sampler {
  std::vector<std::string> pw, fq;
  std::atomic<bool> on{false}, stop{false};
  std::vector<double> psum, fsum; long n = 0;
  std::thread th;
  static double rd(const std::string& p) { FILE* f = fopen(p.c_str(), "r"); if (!f) return -1; double v = -1; if (fscanf(f, "%lf", &v) != 1) v = -1; fclose(f); return v; }
  Sampler() {
    glob_t g;
    for (const char* pat : {"/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"}) {
      if (glob(pat, 0, nullptr, &g) == 0) { for (size_t i = 0; i < g.gl_pathc; ++i) pw.push_back(g.gl_pathv[i]); globfree(&g); }
      if (!pw.empty()) break;
    }
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", 0, nullptr, &g) == 0) { for (size_t i = 0; i < g.gl_pathc; ++i) fq.push_back(g.gl_pathv[i]); globfree(&g); }
    psum.assign(pw.size(), 0); fsum.assign(fq.size(), 0);
    th = std::thread([this] {
      while (!stop) {
        if (on) { for (size_t i = 0; i < pw.size(); ++i) psum[i] += rd(pw[i]); for (size_t i = 0; i < fq.size(); ++i) fsum[i] += rd(fq[i]); ++n; }
        usleep(2000);
      }
    });
  }
  void begin() { for (auto& v : psum) v = 0; for (auto& v : fsum) v = 0; n = 0; on = true; }
  std::string end() {
    on = false; usleep(5000);
    char buf[64]; std::string s;
    if (!n || pw.empty()) return " power: n/a";
    s = " power W:";
    for (size_t i = 0; i < pw.size(); ++i) { snprintf(buf, sizeof buf, " %.0f", psum[i] / n * 1e-6); s += buf; }
    s += "  sclk MHz:";
    for (size_t i = 0; i < fq.size(); ++i) { snprintf(buf, sizeof buf, " %.0f", fsum[i] / n * 1e-6); s += buf; }
    snprintf(buf, sizeof buf, " (%ld samples)", n); s += buf;
    return s;
  }
  ~Sampler() { stop = true; th.join(); }
};

static s16x8 *d_w, *d_a[3];
static float* d_out; static long long* d_st;
static Sampler* smp;

static int g_reps = 6;
template <int SHAPE, int ROLE, int SRC, int PB = 2, int CB = 4, int ORDER = 0>
void run(int data, int wgs, const char* what) {
  const int blocks = 400;                      // the same FLOPs per block for every form: 72 * 8 * 32768 = 36 * 32 * 16384 per wave
  constexpr int KSTEPS = SHAPE == 0 ? 72 : 36 * (32 / (PB * CB));
  auto launch = [&] { hipLaunchKernelGGL((k<SHAPE, ROLE, SRC, PB, CB, ORDER>), dim3(wgs), dim3(256), 0, 0, d_w, d_a[data], d_out, d_st, blocks); };
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  smp->begin();
  const int reps = g_reps;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  std::string pws = smp->end();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(wgs * 2); hipMemcpy(h.data(), d_st, wgs * 16, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0; for (int i = 0; i < wgs; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  const double nm = (double)blocks * KSTEPS * PB * CB;                        // MFMAs per wave and launch
  const double flops = nm * (SHAPE == 0 ? 32768.0 : 16384.0) * 4 * wgs * reps;
  printf("%-9s %s %dx%d%s %s %s wg/CU %d: %7.1f TFLOP/s  %5.1f cycles/MFMA/wave  shader clock %4.0f MHz  %6.2f ms/launch %s\n", what,
         SHAPE == 0 ? "32x32x16" : "16x16x32", PB, CB, ORDER ? " px-major" : "", ROLE == 0 ? "A=w B=act" : "A=act B=w", SRC ? "LDS " : "regs", wgs / 256,
         flops / (ms * 1e-3) / 1e12, cyc / wgs / nm, cyc / wall * 100.0, ms / reps, pws.c_str());
  fflush(stdout);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
  const bool lng = argc > 1 && !strcmp(argv[1], "long");   // long: ~1.5 s per form (the hwmon power average needs about a second), fewer forms
  smp = new Sampler();
  std::vector<unsigned short> hw(NWF * 512), ha(NAF * 512);
  for (auto& v : hw) v = bf16(0.05f * nrand());
  hipMalloc(&d_w, hw.size() * 2); hipMemcpy(d_w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  for (int d = 0; d < 3; ++d) {
    for (auto& v : ha) { const float x = nrand(); v = d == 0 ? 0 : bf16(d == 1 ? (x > 0 ? x : 0.f) : x); }
    hipMalloc(&d_a[d], ha.size() * 2); hipMemcpy(d_a[d], ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  }
  hipMalloc(&d_out, 512 * 256 * 4); hipMalloc(&d_st, 512 * 16);
  const char* names[3] = {"zero", "post-ReLU", "dense"};
  { smp->begin(); usleep(300000); printf("idle: %s\n", smp->end().c_str()); }
  if (lng) {
    g_reps = 150;
    for (int d = 0; d < 3; ++d) {
      run<0, 0, 0>(d, 512, names[d]); run<1, 0, 0>(d, 512, names[d]); run<1, 0, 0, 4, 8>(d, 512, names[d]);
      run<0, 0, 1>(d, 512, names[d]); run<1, 0, 1>(d, 512, names[d]); run<1, 0, 1, 4, 8>(d, 512, names[d]);
      run<0, 0, 1, 2, 4, 1>(d, 512, names[d]); run<1, 0, 1, 4, 8, 1>(d, 512, names[d]); run<1, 0, 0, 4, 8, 1>(d, 512, names[d]);
    }
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    for (int d = 0; d < 3; ++d) {
      run<0, 0, 0>(d, 512, names[d]); run<0, 1, 0>(d, 512, names[d]);
      run<1, 0, 0>(d, 512, names[d]); run<1, 1, 0>(d, 512, names[d]);
      run<0, 0, 1>(d, 512, names[d]); run<0, 1, 1>(d, 512, names[d]);
      run<1, 0, 1>(d, 512, names[d]);
      run<0, 0, 0>(d, 256, names[d]); run<0, 0, 1>(d, 256, names[d]);
    }
  }
  return 0;
}
