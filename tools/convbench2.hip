// Check + microbenchmark of conv_nhwc_kernel (brever_amd/csrc/conv_nhwc.hip):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 [-DCN_ABL=<bits>] tools/convbench2.hip -o /tmp/convbench2
//   /tmp/convbench2 [B]      -- small shapes are compared with a CPU loop, large ones timed
#include "../brever_amd/csrc/conv_nhwc.hip"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float frand(unsigned long long& s) {
  s = s*6364136223846793005ULL + 1442695040888963407ULL;
  return (float)((s >> 33) & 0xffffff)/(float)0x1000000 - 0.5f;
}
static float h16(float v) { return (float)(_Float16)v; }

struct Case { int B, H, W, C1, C1s, C2, C2s, Cout, fold, silu, res, bias; float scale; };

static int run_case(const Case& c, bool check, int iters) {
  const int Cin = c.C1 + c.C2;
  const size_t npx = (size_t)c.B*c.H*c.W;
  std::vector<_Float16> hx1(npx*c.C1s), hx2(c.C2 ? npx*c.C2s : 1), hres(c.res ? npx*c.Cout : 1);
  std::vector<float> hw((size_t)c.Cout*Cin*9), hb(c.Cout), hsc((size_t)c.B*Cin), hsh((size_t)c.B*Cin);
  unsigned long long s = 12345 + c.H*131 + c.W;
  for (size_t i = 0; i < npx; ++i)
    for (int k = 0; k < c.C1s; ++k) hx1[i*c.C1s + k] = (_Float16)(k < c.C1 ? 2.f*frand(s) : 0.f);
  if (c.C2) for (size_t i = 0; i < npx; ++i)
    for (int k = 0; k < c.C2s; ++k) hx2[i*c.C2s + k] = (_Float16)(k < c.C2 ? 2.f*frand(s) : 0.f);
  if (c.res) for (auto& v : hres) v = (_Float16)frand(s);
  for (auto& v : hw) v = frand(s)*0.1f;
  for (auto& v : hb) v = frand(s);
  for (auto& v : hsc) v = 1.f + frand(s);
  for (auto& v : hsh) v = frand(s);
  _Float16 *x1, *x2 = nullptr, *res = nullptr, *y; float *w, *b, *sc, *sh; void* wp;
  CK(hipMalloc(&x1, hx1.size()*2)); CK(hipMemcpy(x1, hx1.data(), hx1.size()*2, hipMemcpyHostToDevice));
  if (c.C2) { CK(hipMalloc(&x2, hx2.size()*2)); CK(hipMemcpy(x2, hx2.data(), hx2.size()*2, hipMemcpyHostToDevice)); }
  if (c.res) { CK(hipMalloc(&res, hres.size()*2)); CK(hipMemcpy(res, hres.data(), hres.size()*2, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&y, npx*c.Cout*2)); CK(hipMemset(y, 0xff, npx*c.Cout*2));
  CK(hipMalloc(&w, hw.size()*4)); CK(hipMemcpy(w, hw.data(), hw.size()*4, hipMemcpyHostToDevice));
  CK(hipMalloc(&b, hb.size()*4)); CK(hipMemcpy(b, hb.data(), hb.size()*4, hipMemcpyHostToDevice));
  CK(hipMalloc(&sc, hsc.size()*4)); CK(hipMemcpy(sc, hsc.data(), hsc.size()*4, hipMemcpyHostToDevice));
  CK(hipMalloc(&sh, hsh.size()*4)); CK(hipMemcpy(sh, hsh.data(), hsh.size()*4, hipMemcpyHostToDevice));
  const long long np = brv_conv_nhwc_packed_size(c.Cout, Cin, 3);
  CK(hipMalloc(&wp, np*2));
  if (brv_conv_nhwc_pack(w, wp, c.Cout, Cin, 3, 0)) { printf("pack failed\n"); exit(1); }
  double* stats = nullptr;
  CK(hipMalloc(&stats, (size_t)c.B*c.Cout*16)); CK(hipMemset(stats, 0, (size_t)c.B*c.Cout*16));
  auto launch = [&]() {
    return brv_conv_nhwc_forward(x1, c.C1, c.C1s, x2, c.C2, c.C2s, wp, c.bias ? b : nullptr, res, c.Cout,
                                 c.fold ? sc : nullptr, c.fold ? sh : nullptr, c.silu, y, c.Cout, c.B,
                                 c.H, c.W, c.Cout, 3, c.scale, stats, 0);
  };
  int rc = launch();
  if (rc) { printf("launch rc %d\n", rc); exit(1); }
  CK(hipDeviceSynchronize());
  int bad = 0;
  if (check) {
    std::vector<_Float16> hy(npx*c.Cout);
    CK(hipMemcpy(hy.data(), y, hy.size()*2, hipMemcpyDeviceToHost));
    // activated input, rounded to fp16 like the LDS image
    std::vector<float> act(npx*Cin);
    for (int bb = 0; bb < c.B; ++bb)
      for (size_t i = 0; i < (size_t)c.H*c.W; ++i)
        for (int k = 0; k < Cin; ++k) {
          const size_t pi = (size_t)bb*c.H*c.W + i;
          float v = k < c.C1 ? (float)hx1[pi*c.C1s + k] : (float)hx2[pi*c.C2s + (k - c.C1)];
          if (c.fold) {
            v = hsc[(size_t)bb*Cin + k]*v + hsh[(size_t)bb*Cin + k];
            if (c.silu) v = v/(1.f + expf(-v));
            v = h16(v);
          }
          act[pi*Cin + k] = v;
        }
    std::vector<double> hst((size_t)c.B*c.Cout*2);
    CK(hipMemcpy(hst.data(), stats, hst.size()*8, hipMemcpyDeviceToHost));
    {
      double werr = 0.0;
      for (int bb = 0; bb < c.B; ++bb)
        for (int co = 0; co < c.Cout; ++co) {
          double s1 = 0.0, s2 = 0.0;
          for (size_t i = 0; i < (size_t)c.H*c.W; ++i) { const double v = (float)hy[((size_t)bb*c.H*c.W + i)*c.Cout + co]; s1 += v; s2 += v*v; }
          const double e1 = fabs(hst[((size_t)bb*c.Cout + co)*2] - s1)/(1.0 + fabs(s1)), e2 = fabs(hst[((size_t)bb*c.Cout + co)*2 + 1] - s2)/(1.0 + s2);
          if (e1 > werr) werr = e1; if (e2 > werr) werr = e2;
        }
      printf("  stats worst rel err %.2e\n", werr);
      if (!(werr < 1e-4)) ++bad;
    }
    double worst = 0.0;
    std::vector<int> badmap((size_t)c.B*c.H*c.W, 0);
    for (int bb = 0; bb < c.B; ++bb)
      for (int h = 0; h < c.H; ++h)
        for (int ww = 0; ww < c.W; ++ww)
          for (int co = 0; co < c.Cout; ++co) {
            double a = c.bias ? hb[co] : 0.0;
            for (int kh = 0; kh < 3; ++kh)
              for (int kw = 0; kw < 3; ++kw) {
                const int hh = h + kh - 1, w2 = ww + kw - 1;
                if (hh < 0 || hh >= c.H || w2 < 0 || w2 >= c.W) continue;
                const float* ap = &act[(((size_t)bb*c.H + hh)*c.W + w2)*Cin];
                for (int k = 0; k < Cin; ++k) a += (double)h16(hw[((size_t)co*Cin + k)*9 + kh*3 + kw])*ap[k];
              }
            const size_t pi = ((size_t)bb*c.H + h)*c.W + ww;
            if (c.res) a += (float)hres[pi*c.Cout + co];
            a *= c.scale;
            const double got = (float)hy[pi*c.Cout + co];
            const double err = fabs(got - a)/(1.0 + fabs(a));
            if (err > worst) worst = err;
            if (!(err < 4e-3)) { if (bad < 5) printf("  mismatch b%d h%d w%d co%d: got %g want %g\n", bb, h, ww, co, got, a); ++bad; ++badmap[pi]; }
          }
    if (bad && c.W <= 80) {
      for (int h = 0; h < c.H; ++h) {
        for (int ww = 0; ww < c.W; ++ww) { const int n = badmap[(size_t)h*c.W + ww]; putchar(n == 0 ? '.' : n < c.Cout/2 ? 'o' : 'X'); }
        putchar('\n');
      }
    }
    printf("check B%d %dx%d %d(+%d)->%d fold%d silu%d res%d: worst %.2e, %d bad\n", c.B, c.H, c.W, c.C1, c.C2, c.Cout, c.fold, c.silu, c.res, worst, bad);
  } else {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms/iters*1e3, fl = 2.0*c.B*Cin*c.Cout*9.0*c.H*c.W;
#ifdef CN_DIAG
    {
      std::vector<unsigned long long> d(512);
      CK(hipMemcpy(d.data(), brv_conv_nhwc_dbg, d.size()*8, hipMemcpyDeviceToHost));
      printf("kernel span: %llu cycles, %llu realtime ticks (100 MHz) -> %.2f GHz\n", d[288], d[289], d[288]/(d[289]*10.0));
      for (int wv = 0; wv < 8; ++wv) {
        printf("wave %d:", wv);
        for (int a = 0; a < 9; ++a) {
          const unsigned long long* q = &d[(wv*9 + a)*4];
          const unsigned long long prev3 = a ? d[(wv*9 + a - 1)*4 + 3] : q[0];
          printf(" | dma%4llu vm%4llu bar%4llu mfma%4llu", q[0] - prev3, q[1] - q[0], q[2] - q[1], q[3] - q[2]);
        }
        printf("\n");
      }
      for (int wv = 0; wv < 8; wv += 4) {
        printf("epilogue wave %d:", wv);
        const unsigned long long* e = &d[300 + wv*13];
        for (int a = 1; a < 13; ++a) printf(" %s%llu", a % 3 == 1 ? "| wr+bar " : a % 3 == 2 ? "rd " : "st ", e[a] - e[a - 1]);
        printf("\n");
      }
    }
#endif
    printf("time abl %2d B%d %dx%d %d(+%d)->%d fold%d res%d: %8.1f us %7.1f TFLOP/s (%.1f%% of 2.5 PF)\n", CN_ABL, c.B, c.H, c.W, c.C1, c.C2,
           c.Cout, c.fold, c.res, us, fl/us/1e6, fl/us/1e6/25.0);
  }
  hipFree(stats); hipFree(x1); if (x2) hipFree(x2); if (res) hipFree(res); hipFree(y); hipFree(w); hipFree(b); hipFree(sc); hipFree(sh); hipFree(wp);
  return bad;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1;
  int bad = 0;
  if (!CN_ABL) {
    const Case checks[] = {
      {1, 16, 32, 32, 32, 0, 0, 128, 0, 0, 0, 0, 1.f},
      {2, 20, 37, 32, 32, 0, 0, 40, 0, 0, 0, 1, 1.f},
      {1, 35, 70, 64, 64, 0, 0, 128, 1, 1, 1, 1, 0.70710678f},
      {2, 17, 33, 64, 64, 32, 32, 136, 1, 0, 0, 1, 1.f},
      {1, 40, 45, 4, 8, 0, 0, 128, 0, 0, 0, 1, 1.f},
      {3, 8, 16, 96, 96, 64, 64, 256, 1, 1, 1, 1, 0.5f},
      {1, 64, 126, 128, 128, 0, 0, 128, 1, 1, 1, 1, 1.f},
    };
    for (const Case& c : checks) bad += run_case(c, true, 0);
  }
  const Case times[] = {
    {B, 256, 501, 128, 128, 0, 0, 128, 0, 0, 0, 1, 1.f},
    {B, 256, 501, 128, 128, 0, 0, 128, 1, 1, 1, 1, 1.f},
    {B, 256, 512, 128, 128, 0, 0, 128, 1, 1, 1, 1, 1.f},
    {B, 256, 501, 128, 128, 128, 128, 128, 1, 1, 0, 1, 1.f},
    {B, 128, 251, 128, 128, 0, 0, 128, 1, 1, 1, 1, 1.f},
    {B, 64, 126, 256, 256, 0, 0, 256, 1, 1, 1, 1, 1.f},
    {B, 32, 63, 256, 256, 0, 0, 256, 1, 1, 1, 1, 1.f},
  };
  for (const Case& c : times) run_case(c, false, 20);
  printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
