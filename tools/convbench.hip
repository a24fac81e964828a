// Microbenchmark of conv_mfma_kernel (brever_amd/csrc/conv_mfma.hip) with compile-time
// ablations: hipcc -DCM_ABL=<bits> tools/convbench.hip -o tools/convbench_<bits>
#include "../brever_amd/csrc/conv_mfma.hip"
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1;
  struct S { int ci, co, k, H, W; };
  const S shapes[] = {{128, 128, 3, 256, 501}, {128, 128, 3, 256, 512}, {128, 128, 3, 256, 480}, {256, 128, 3, 256, 501}, {256, 256, 3, 64, 126}, {256, 256, 3, 4, 8}};
  for (const S& s : shapes) {
    float *x, *y, *w, *bias; void* wp;
    const size_t nx1 = (size_t)s.ci*s.H*s.W, ny1 = (size_t)s.co*s.H*s.W, nx = nx1*B, ny = ny1*B, nw = (size_t)s.co*s.ci*s.k*s.k;
    CK(hipMalloc(&x, nx*4)); CK(hipMalloc(&y, ny*4)); CK(hipMalloc(&w, nw*4)); CK(hipMalloc(&bias, s.co*4));
    std::vector<float> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = (float)((i*2654435761u) % 1000)/1000.f - 0.5f;
    for (size_t i = 0; i < nw; ++i) hw[i] = ((float)((i*40503u) % 1000)/1000.f - 0.5f)*0.05f;
    CK(hipMemcpy(x, hx.data(), nx*4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), nw*4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, s.co*4));
    CK(hipMalloc(&wp, brv_conv2d_packed_size(s.co, s.ci, s.k)*2));
    brv_conv2d_pack_f16(w, wp, s.co, s.ci, s.k, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) brv_conv2d_mfma_forward(x, wp, bias, nullptr, nullptr, nullptr, 0, y, B, s.ci, s.H, s.W, s.co, s.k, nx1, ny1, 1.f, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) brv_conv2d_mfma_forward(x, wp, bias, nullptr, nullptr, nullptr, 0, y, B, s.ci, s.H, s.W, s.co, s.k, nx1, ny1, 1.f, 0);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms/20*1e3, fl = 2.0*B*s.ci*s.co*s.k*s.k*s.H*s.W;
    printf("B %d abl %2d ring %d: %3d->%3d k%d %3dx%3d: %8.1f us %7.1f TFLOP/s\n", B, CM_ABL, CM_RING, s.ci, s.co, s.k, s.H, s.W, us, fl/us/1e6);
    hipFree(x); hipFree(y); hipFree(w); hipFree(bias); hipFree(wp);
  }
  return 0;
}
