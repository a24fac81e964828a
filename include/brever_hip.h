/* brever_hip.h -- C ABI of the MI355X (gfx950) hot path of brever_amd.
 *
 * The reference (philgzl/brever) is pure Python/PyTorch and has no FFI; its
 * boundary for this path is the Python plugin API (brever/models/base.py:12-358,
 * brever/criterion.py, brever/models/convtasnet/convtasnet.py). This header is
 * the C boundary the MI355X build puts *underneath* that API: each entry point
 * names the reference code it replaces. Conventions:
 *
 *   - every pointer is a device pointer borrowed from the caller (PyTorch's
 *     allocator); the library allocates nothing and keeps no global state;
 *   - every call takes the HIP stream to launch on and never synchronises;
 *   - return value: 0 ok, < 0 invalid argument / unsupported configuration,
 *     > 0 a hipError_t; brv_last_error() gives a thread-local message.
 *
 * Activations inside the library are channels-last bf16 ([item][frame][channel],
 * channels padded to a multiple of 64); this never leaks through the ABI: inputs
 * and outputs are the reference's fp32 (batch, [sources,] samples) tensors and
 * the parameters are one flat fp32 buffer in the reference's parameter order.
 */
#ifndef BREVER_HIP_H
#define BREVER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* brv_stream_t;          /* hipStream_t */

int brv_version(void);
const char* brv_last_error(void);

/* Optional per-launch timing with HIP events on the launch stream (bench.py's roofline
 * measurement). A profiler is an object the CALLER owns: launches made with
 * brv_launch_opts.prof = handle are bracketed by two events each; brv_prof_collect writes
 * "label calls total_ms flops bytes" lines (algorithmic FLOPs / HBM bytes as stated in DESIGN.md)
 * for them, returns the buffer size needed and clears the handle's records. by_dilation != 0
 * labels the depthwise backward kernels per dilation. One handle per thread of calls. */
void* brv_prof_create(int by_dilation);
int64_t brv_prof_collect(void* prof, char* buf, int64_t buflen);
void brv_prof_destroy(void* prof);

/* Per-call launch options of the Conv-TasNet entry points (brv_ctn_prepare / _forward /
 * _backward / _backward_part). NULL or a zeroed struct = the measured-best path on the whole
 * chip. The library keeps NO process-global state (SURVEY.md 8b "Threading / streams"): what used
 * to be a global switch (the share of the chip a kernel chain takes), a global profiler and
 * environment reads inside the library is an argument, so two models may step from two host
 * threads on two streams. The switches exist for A/B measurements and tests; the
 * Python host fills them from the environment (brever_amd/hip.py: launch_opts). */
#define BRV_OPT_NO_FWD_FUSE     0x001u  /* forward as three launches per block instead of the fused stage */
#define BRV_OPT_NO_BWD_FUSE     0x002u  /* backward as three launches per block (bwd_fused.cuh off) */
#define BRV_OPT_NO_WS           0x004u  /* generic tile GEMM instead of the persistent kernels */
#define BRV_OPT_DWPW2_WS        0x008u  /* fused forward stage as a mode of the persistent GEMM */
#define BRV_OPT_NO_DZ_FUSE      0x010u  /* gLN_2 / PReLU_2 backward as a pass of its own */
#define BRV_OPT_NO_DZ1_FUSE     0x020u  /* gLN_1 / PReLU_1 backward as a pass of its own */
#define BRV_OPT_NO_WGRAD_FULL   0x040u  /* grouped generic kernel for the [res | skip] weight gradient */
#define BRV_OPT_NO_WGRAD_SPLIT  0x080u  /* no item split of that launch */
#define BRV_OPT_NO_PW1_RC       0x100u  /* first-conv data gradient reads the stored z1 (pw1_bwd.cuh off) */
#define BRV_OPT_PW1_RC_WGRAD    0x200u  /* first-conv weight gradient rebuilds dz1 from e1 and x too: dz1 is never stored */
#define BRV_OPT_BWD_PERSIST    0x2000u  /* fused backward stage in its persistent form (bwd_fused_p.cuh: two tiles per workgroup; measured slower) */
#define BRV_OPT_NO_WGRAD_128   0x1000u  /* [res | skip] weight gradient with 64 instead of 128 H channels per workgroup (gemm_wgrad_full.cuh) */
#define BRV_OPT_DWPW2_V2        0x800u  /* fused forward stage in its whole-row form (dwpw2_fused_v2.cuh) */
#define BRV_OPT_PW1_RC_TILES    0x400u  /* recompute in the one-tile-per-workgroup form (weights re-read per tile; implies _WGRAD) */
typedef struct brv_launch_opts {
  uint32_t size;          /* sizeof(brv_launch_opts), for forward compatibility */
  uint32_t flags;         /* BRV_OPT_* */
  int32_t cu_eighths;     /* persistent kernels launch cu_eighths/8 of one workgroup per CU; 0 = 8 = all.
                             The host passes 7 while two half-batch chains share the chip */
  int32_t wg_target;      /* workgroups a weight-gradient launch aims for; 0 = per-shape defaults */
  void* prof;             /* brv_prof_create handle or NULL */
} brv_launch_opts;

/* ---- Conv-TasNet ---------------------------------------------------------
 * Hyper-parameters of brever.models.convtasnet.ConvTasNet.__init__
 * (convtasnet.py:30-46). causal != 0 selects the cumulative layer norm and all-left depthwise padding
 * (convtasnet.py:244-247, modules/normalization.py:5-62). */
typedef struct brv_ctn_config {
  int32_t filters, filter_length, bottleneck_channels, hidden_channels,
          skip_channels, kernel_size, layers, repeats, output_sources, causal;
} brv_ctn_config;

/* Number of fp32 parameters, in the order of ConvTasNet.parameters()
 * (SURVEY.md App. A.3 "init-order contract"); 4 935 217 at the defaults. */
int64_t brv_ctn_param_count(const brv_ctn_config* cfg);
/* Offset (in floats) of the i-th parameter tensor and their number (343). */
int64_t brv_ctn_param_tensors(const brv_ctn_config* cfg);
int64_t brv_ctn_param_offset(const brv_ctn_config* cfg, int64_t index);

/* Frames produced by Encoder.pad + Conv1d (convtasnet.py:115-126). */
int64_t brv_ctn_frames(const brv_ctn_config* cfg, int64_t length);

/* Bytes of the prepared-weight buffer (bf16 GEMM operands, both layouts). */
int64_t brv_ctn_prepared_bytes(const brv_ctn_config* cfg);
/* Bytes of the activation workspace for a (batch, length) input. The saved
 * activations live here between forward and backward. */
int64_t brv_ctn_workspace_bytes(const brv_ctn_config* cfg, int64_t batch,
                                int64_t length);
/* Byte offset of a named workspace tensor ("w", "x", "z1", "z2", "skip", "m",
 * "y", "stats"; index selects the block) -- for tests and profiling only. */
int64_t brv_ctn_workspace_offset(const brv_ctn_config* cfg, int64_t batch,
                                 int64_t length, const char* name, int64_t index);

/* fp32 parameters -> prepared bf16 operands. Call after every parameter update (with the options
 * the forward / backward calls will get: BRV_OPT_NO_FWD_FUSE selects which [res | skip] operand
 * forms are prepared). */
int brv_ctn_prepare(const brv_ctn_config* cfg, const float* params,
                    void* prepared, const brv_launch_opts* opts, brv_stream_t stream);

/* ConvTasNet.forward (convtasnet.py:66-72): wave (batch, length) fp32, rows `wave_stride` floats
 * apart (0 = length; the trainer's batch[:, 0] of a (batch, 1 + sources, length) tensor is read in
 * place with wave_stride = (1 + sources)*length) -> out (batch, sources, length) fp32. Leaves the
 * activations needed by brv_ctn_backward in `workspace`. */
int brv_ctn_forward(const brv_ctn_config* cfg, const float* params,
                    const void* prepared, void* workspace, const float* wave,
                    int64_t wave_stride, float* out, int64_t batch, int64_t length,
                    const brv_launch_opts* opts, brv_stream_t stream);

/* Autograd of ConvTasNet.forward: d_out (batch, sources, length) fp32 ->
 * gradients ACCUMULATED into `grads` (flat fp32, same layout as params;
 * the caller zeroes it, as optimizer.zero_grad() does). */
int brv_ctn_backward(const brv_ctn_config* cfg, const float* params,
                     const void* prepared, void* workspace, const float* wave,
                     int64_t wave_stride, const float* d_out, float* grads, int64_t batch,
                     int64_t length, const brv_launch_opts* opts, brv_stream_t stream);

/* The backward pass in `nparts` parts, for overlapping the data-parallel gradient all-reduce
 * with the rest of backward (the reference wraps the model in
 * DistributedDataParallel, brever/training.py:62-63, whose reducer buckets gradients the same
 * way). Part 0 runs the decoder / mask / output-conv gradients and the LAST group of TCN blocks,
 * the last part the first group and the bottleneck / encoder; each part finishes ALL gradients
 * (data, weight, per-channel) of its blocks. brv_ctn_grad_bucket gives the contiguous range of
 * the flat gradient that is final once part `part` has run (the ranges of all parts tile the
 * buffer): the caller may all-reduce that slice on another stream while later parts compute.
 * Calling parts 0 .. nparts-1 in order equals one brv_ctn_backward call. */
int brv_ctn_backward_part(const brv_ctn_config* cfg, const float* params,
                          const void* prepared, void* workspace, const float* wave,
                          int64_t wave_stride, const float* d_out, float* grads, int64_t batch,
                          int64_t length, int32_t part, int32_t nparts,
                          const brv_launch_opts* opts, brv_stream_t stream);
int brv_ctn_grad_bucket(const brv_ctn_config* cfg, int32_t part, int32_t nparts,
                        int64_t* offset, int64_t* count);

/* The same model with fp32 activations and products of fp32 accuracy (the fp32 MFMA, or -- long
 * products contiguous in the reduction index -- three bf16 pieces per fp32 operand and six bf16
 * MFMAs with fp32 accumulation: csrc/gemm_f32_big.hip; every sum in a fixed order, so results are
 * bitwise repeatable): ConvTasNet.forward WITHOUT
 * autocast (convtasnet.py:78-97 with use_amp=False -- `enhance(x, use_amp=False)` of
 * scripts/test_model.py:173-175, BreverTrainer(use_amp=False)). No prepared operands: the flat
 * fp32 parameters are read directly. Own workspace layout (brv_ctn_f32_workspace_bytes);
 * causal and non-causal; gradients are ACCUMULATED into `grads` like brv_ctn_backward. */
int64_t brv_ctn_f32_workspace_bytes(const brv_ctn_config* cfg, int64_t batch,
                                    int64_t length);
int brv_ctn_f32_forward(const brv_ctn_config* cfg, const float* params,
                        void* workspace, const float* wave, float* out,
                        int64_t batch, int64_t length, brv_stream_t stream);
int brv_ctn_f32_backward(const brv_ctn_config* cfg, const float* params,
                         void* workspace, const float* wave, const float* d_out,
                         float* grads, int64_t batch, int64_t length,
                         brv_stream_t stream);
int brv_ctn_f32_backward_part(const brv_ctn_config* cfg, const float* params,
                              void* workspace, const float* wave, const float* d_out,
                              float* grads, int64_t batch, int64_t length,
                              int32_t part, int32_t nparts, brv_stream_t stream);

/* ---- criteria (brever/criterion.py) ---------------------------------------
 * x, y: (batch, sources, length) fp32 contiguous rows with `stride` floats
 * between rows; lengths: (batch,) int64 on the device; scratch: at least
 * brv_loss_scratch_bytes(batch, sources) bytes; loss: (batch,) fp32. */
int64_t brv_loss_scratch_bytes(int64_t batch, int64_t sources);

/* snr (criterion.py:75-101). coef (batch*sources) is kept in scratch for bwd. */
int brv_snr_forward(const float* x, const float* y, const int64_t* lengths,
                    int64_t batch, int64_t sources, int64_t length,
                    int64_t stride, void* scratch, float* loss,
                    brv_stream_t stream);
/* d loss[b] * gscale[b] / d x -> dx (same layout as x). */
int brv_snr_backward(const float* x, const float* y, const int64_t* lengths,
                     int64_t batch, int64_t sources, int64_t length,
                     int64_t stride, const void* scratch, const float* gscale,
                     float* dx, brv_stream_t stream);
/* The same with the labels read in place from the trainer's (batch, 1 + sources, length) tensor:
 * row (item b, source s) of y starts at y + b*y_batch_stride + s*y_source_stride (no copy of
 * batch[:, 1:], which was a strided torch copy kernel per step). */
int brv_snr_forward_strided(const float* x, const float* y, int64_t y_batch_stride,
                            int64_t y_source_stride, const int64_t* lengths, int64_t batch,
                            int64_t sources, int64_t length, int64_t stride, void* scratch,
                            float* loss, brv_stream_t stream);
int brv_snr_backward_strided(const float* x, const float* y, int64_t y_batch_stride,
                             int64_t y_source_stride, const int64_t* lengths, int64_t batch,
                             int64_t sources, int64_t length, int64_t stride, const void* scratch,
                             const float* gscale, float* dx, brv_stream_t stream);
/* sisnr with PIT (criterion.py:21-72), sources <= 4. The forward leaves the winning
 * permutation and the gradient coefficients in scratch for the backward. */
int brv_sisnr_forward(const float* x, const float* y, const int64_t* lengths,
                      int64_t batch, int64_t sources, int64_t length,
                      int64_t stride, void* scratch, float* loss,
                      brv_stream_t stream);
int brv_sisnr_backward(const float* x, const float* y, const int64_t* lengths,
                       int64_t batch, int64_t sources, int64_t length,
                       int64_t stride, const void* scratch, const float* gscale,
                       float* dx, brv_stream_t stream);
/* mse (criterion.py:104-132) on real tensors; weight may be NULL. */
int brv_mse_forward(const float* x, const float* y, const int64_t* lengths,
                    const float* weight, int64_t batch, int64_t sources,
                    int64_t length, int64_t stride, void* scratch, float* loss,
                    brv_stream_t stream);
int brv_mse_backward(const float* x, const float* y, const int64_t* lengths,
                     const float* weight, int64_t batch, int64_t sources,
                     int64_t length, int64_t stride, const float* gscale,
                     float* dx, brv_stream_t stream);

/* ---- STFT / iSTFT / filterbank (brever/modules/stft.py) ------------------------
 * One-sided transforms, hop dividing the frame length, center=True with constant
 * (zero) padding exactly as STFT.forward / STFT.backward of the reference
 * (stft.py:59-149). The DFT bases are fp32 matrices built by the caller:
 *   basis     [2*(n/2+1)][n]  row 2k = w[m] cos(2 pi k m/n)*norm, row 2k+1 = -w[m] sin(...)*norm
 *   inv_basis [n][2*(n/2+1)]  windowed inverse real DFT (norm undone)
 * spec is complex64 laid out (rows, n/2+1, frames) like torch.stft's output. */
int64_t brv_stft_frames(int64_t length, int64_t frame_length, int64_t hop_length);
int brv_stft_forward(const float* x, const float* basis, float* spec, int64_t rows,
                     int64_t length, int64_t frame_length, int64_t hop_length,
                     float compression, float scale, brv_stream_t stream);
/* frames_scratch: rows*frames*frame_length floats; y: (rows, hop*(frames-1)). */
int brv_istft_backward(const float* spec, const float* inv_basis, const float* window,
                       float* frames_scratch, float* y, int64_t rows, int64_t frames,
                       int64_t frame_length, int64_t hop_length, float compression,
                       float scale, brv_stream_t stream);
/* Adjoint of brv_stft_forward wrt x for compression == 1 (autograd of STFT.forward,
 * stft.py:59-99): dx (rows, length) from dspec (rows, n/2+1, frames) complex64;
 * frames_scratch: rows*frames*frame_length floats. */
int brv_stft_adjoint(const float* dspec, const float* basis, float* frames_scratch, float* dx,
                     int64_t rows, int64_t length, int64_t frame_length, int64_t hop_length,
                     float scale, brv_stream_t stream);
/* Framed DFT with explicit geometry and its transpose (ConvSTFT.forward / backward,
 * stft.py:201-319: STFT as a strided convolution / transposed convolution with the same
 * filters): frame t covers samples [t*hop - pad_left, ... + n); basis rows interleave
 * (real_k, imag_k) like brv_stft_forward's. */
int brv_framed_dft_forward(const float* x, const float* basis, float* spec, int64_t rows,
                           int64_t length, int64_t frame_length, int64_t hop_length,
                           int64_t pad_left, int64_t frames, float compression, float scale,
                           brv_stream_t stream);
int brv_framed_dft_transpose(const float* spec, const float* basis, float* frames_scratch,
                             float* y, int64_t rows, int64_t frames, int64_t frame_length,
                             int64_t hop_length, int64_t pad_left, int64_t out_len,
                             float compression, float scale, brv_stream_t stream);
/* The transforms of the STFT module in full generality (stft.py:32-149: any hop, n_fft !=
 * frame_length through a centrally zero-padded window, one- or two-sided spectra) and at FFT
 * accuracy: the DFT basis is passed in DOUBLE precision and the product accumulates on the fp64
 * matrix pipe (v_mfma_f64_16x16x4_f64); data stay fp32 in memory.
 *   dft64_forward:   spec (rows, bins, frames) complex64 = framed DFT of x (rows, length); frame t
 *                    covers samples [t*hop - pad_left, + n) (zeros outside); basis (2 bins, n)
 *                    rows (re_k, im_k); then |.|^compression e^{j angle} * scale.
 *   dft64_synthesis: frames_out (rows, frames, n) = (spec / scale, decompressed)^T x tbasis
 *                    (2 bins, n): inverse transform (tbasis = inverse basis^T) or adjoint of the
 *                    forward transform (tbasis = forward basis).
 *   overlap_add:     y (rows, out_len) from frames (rows, frames, n); window != NULL divides by
 *                    the window-square envelope (torch.istft).
 *   spec_compress(_backward): Y = scale |X|^(c-1) X on n complex64 values and its gradient. */
int brv_dft64_forward(const float* x, const double* basis, float* spec, int64_t rows,
                      int64_t length, int64_t n, int64_t hop, int64_t pad_left, int64_t frames,
                      int64_t bins, float compression, float scale, brv_stream_t stream);
int brv_dft64_synthesis(const float* spec, const double* tbasis, float* frames_out, int64_t rows,
                        int64_t frames, int64_t n, int64_t bins, float compression, float scale,
                        brv_stream_t stream);
int brv_overlap_add(const float* frames_in, const float* window, float* y, int64_t rows,
                    int64_t frames, int64_t n, int64_t hop, int64_t pad_left, int64_t out_len,
                    brv_stream_t stream);
int brv_spec_compress(const float* x, float* y, int64_t n, float compression, float scale,
                      brv_stream_t stream);
/* y (rows, out_len) = x (rows, length) shifted right by `left` and extended by mode 0 zeros,
 * 1 reflect, 2 replicate, 3 circular: F.pad of STFT.pad and torch.stft's centre padding for
 * pad_mode != 'constant' (stft.py:140-149,66-77). */
int brv_pad_signal(const float* x, float* y, int64_t rows, int64_t length, int64_t left,
                   int64_t out_len, int mode, brv_stream_t stream);
/* out = mag e^{j phase} / (mag, phase) of n complex64 values ('mag_phase' of stft.py:93-110). */
int brv_polar(const float* mag, const float* phase, float* out, int64_t n, brv_stream_t stream);
int brv_mag_phase(const float* x, float* mag, float* phase, int64_t n, brv_stream_t stream);
int brv_spec_compress_backward(const float* x, const float* gy, float* gx, int64_t n,
                               float compression, float scale, brv_stream_t stream);
/* d[b] = a[b or shared] (M x K) @ b[b] (K x N), fp32 (MelFilterbank.forward/backward,
 * stft.py:189-198). a_batch_stride = 0 shares one matrix across the batch. */
int brv_matmul_f32(const float* a, const float* b, float* d, int64_t batch, int64_t M,
                   int64_t N, int64_t K, int64_t a_batch_stride, brv_stream_t stream);

/* General fp32 GEMM at fp32 accuracy (nn.Linear of the FFNN model and its gradients,
 * models/ffnn/ffnn.py:151-171). 16-byte aligned operands of a product that fills the chip with
 * 256 x 128 tiles run on csrc/gemm_f32_big.hip: the fp32 MFMA, or, for a row-major and b stored N x K
 * with >= 192 tiles, the split-bf16 form (x = hi + mid + lo in bf16, six MFMAs, error below fp32
 * rounding of the products). Other calls run on the 128 x 128 fp32-MFMA kernel; long reductions over few
 * tiles are split there over workgroups that add with atomics (order of arrival: not bitwise
 * repeatable; brv_ctn_f32_* keeps its own weight gradients on a fixed-order split).
 * d[z] (M x N) (+)= sum_kb op_a(a[z,kb]) @ op_b(b[z,kb]) +
 * row_bias[m]; trans_a: a stored (K x M); trans_b: b stored (N x K); kbatch extends the
 * reduction over kbatch operand pairs a/b_kbatch_stride apart (weight gradients summed over
 * the batch). accumulate: 0 overwrite, 1 add to d, 2 overwrite with row_bias read per output
 * COLUMN (bias[n]: nn.Linear on row-major activations without a transposed copy). */
int brv_gemm_f32(const float* a, const float* b, float* d, int64_t batch, int64_t M, int64_t N,
                 int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int64_t a_batch_stride,
                 int64_t b_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                 int64_t kbatch, int64_t a_kbatch_stride, int64_t b_kbatch_stride,
                 const float* row_bias, int accumulate, brv_stream_t stream);
/* brv_gemm_f32 with caller-provided scratch (16-byte aligned, brv_gemm_f32_workspace_bytes of the same shape;
 * 0 = the shape needs none): a long reduction over few output tiles -- the weight gradients of the fp32
 * convolutions (reference: autograd of nn.Conv2d / nn.ConvTranspose2d, models/dccrn/dccrn.py:225-292) -- is
 * split over workgroups whose partial tiles are summed IN SPLIT ORDER by a second kernel (bitwise repeatable,
 * no atomics), and with a row-major / b stored N x K such a product runs in the split-bf16 form (fp32 accuracy
 * at 2.7x the rate of the fp32 MFMA). workspace == NULL: exactly brv_gemm_f32. */
int64_t brv_gemm_f32_workspace_bytes(int64_t batch, int64_t M, int64_t N, int64_t K, int trans_a, int trans_b,
                                     int64_t kbatch);
int brv_gemm_f32_ws(const float* a, const float* b, float* d, int64_t batch, int64_t M, int64_t N,
                    int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int64_t a_batch_stride,
                    int64_t b_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                    int64_t kbatch, int64_t a_kbatch_stride, int64_t b_kbatch_stride,
                    const float* row_bias, int accumulate, float* workspace, int64_t workspace_bytes,
                    brv_stream_t stream);
/* The same product with the operands rounded to bf16 on their way into LDS and fp32
 * accumulation (v_mfma_f32_32x32x16_bf16): the use_amp path of DCCRN's convolutions, LSTM
 * projections and Linear layers (the reference autocasts them, models/dccrn/dccrn.py:113-121). */
int brv_gemm_bf16(const float* a, const float* b, float* d, int64_t batch, int64_t M, int64_t N,
                  int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int64_t a_batch_stride,
                  int64_t b_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                  int64_t kbatch, int64_t a_kbatch_stride, int64_t b_kbatch_stride,
                  const float* row_bias, int accumulate, brv_stream_t stream);
/* brv_gemm_bf16 with op_b = the COLUMN MATRIX of an image that is never written out (implicit GEMM
 * of the DCCRN convolutions, reference brever/models/dccrn/dccrn.py:221-290): rows (c, i, j) of a
 * kh x kw window, columns = the pixels (y, x) of an Ho x Wo grid; mode 1: image[c][y*sh - ph + i]
 * [x*sw - pw + j] (im2col: convolution forward / weight gradient, transposed-convolution data
 * gradient); mode 2: image[c][(y + ph - i)/sh][(x + pw - j)/sw] where divisible (the gather form
 * of col2im: convolution data gradient, transposed-convolution forward); zero outside. image:
 * (batch | kbatch, C, H, W) fp32 with the given strides. */
int brv_gemm_bf16_conv(const float* a, const float* image, float* d, int64_t batch, int64_t M,
                       int64_t N, int64_t K, int64_t lda, int64_t ldd, int64_t a_batch_stride,
                       int64_t image_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                       int64_t kbatch, int64_t a_kbatch_stride, int64_t image_kbatch_stride,
                       const float* row_bias, int accumulate, int mode, int64_t C, int64_t H,
                       int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph,
                       int64_t pw, int64_t Ho, int64_t Wo, brv_stream_t stream);
/* The DCCRN convolutions themselves (ComplexWrapper(nn.Conv2d | nn.ConvTranspose2d) with the
 * model's fixed geometry kernel (5, 2), stride (2, 1), padding (2, 0), output_padding (1, 0):
 * reference brever/models/dccrn/dccrn.py:225-235, 238-292, config/models/dccrn.yaml) as one
 * launch on fp32 (B, C, Hin, Win) images: no column matrix, no scatter pass (csrc/cconv.hip).
 *   transposed = 0: out[m][r][w] = bias[m] + sum_{c,i,j} W[m][c][i][j] in[c][2r - 2 + i][w + j],
 *                   out (B, M, Hin/2, Win - 1) (Hin even): Conv2d forward, ConvTranspose2d data gradient;
 *   transposed = 1: out[m][r][w] = bias[m] + sum_{c,j, i = r mod 2} W[m][c][i][j] in[c][(r + 2 - i)/2][w - j],
 *                   out (B, M, 2 Hin, Win + 1): ConvTranspose2d forward, Conv2d data gradient.
 * Rows / frames outside the image read as zero. Operands are rounded to bf16, sums are fp32 (the
 * use_amp path). brv_cconv_pack turns a fp32 weight matrix with W[m][c][i][j] = wc[m*m_stride +
 * c*c_stride + 2 i + j] into the MFMA operand fragments brv_cconv_rows reads (buffer of
 * brv_cconv_packed_bytes(M, C) bytes); bias may be NULL. */
int64_t brv_cconv_packed_bytes(int64_t M, int64_t C);
int brv_cconv_pack(const float* wc, void* wp, int64_t M, int64_t C, int64_t m_stride, int64_t c_stride,
                   brv_stream_t stream);
/* brv_complex_weight_pack + brv_complex_bias_pack + up to two brv_cconv_pack readings of the packed matrix (the
 * forward's and the data gradient's) in ONE launch: wc (2R x 2C), bias (2 Cb), wp1 (M1, C1, strides) and, when wp2 is
 * not NULL, wp2 (M2, C2, strides) -- the same values the separate calls produce. */
int brv_cconv_pack_complex(const float* wr, const float* wi, const float* br, const float* bi, int64_t R, int64_t C,
                           int64_t Cb, float sign, float* wc, float* bias, void* wp1, int64_t M1, int64_t C1,
                           int64_t m_stride1, int64_t c_stride1, void* wp2, int64_t M2, int64_t C2,
                           int64_t m_stride2, int64_t c_stride2, brv_stream_t stream);
/* in_seg > 0: the input is the channel concatenation [in[:, :seg], in2[:, :seg], in[:, seg:], in2[:, seg:]] of two
 * (B, 2 seg, Hin, Win) tensors (the decoder's skip concatenation torch.cat([real, skip_real, imag, skip_imag]),
 * dccrn.py:213-217, never materialised), C = 4 seg, seg a multiple of 8; out_seg > 0: the output channels are dealt
 * the same way to out and out2 (the gradient of that concatenation), M = 4 seg. 0: in2 / out2 are ignored. */
int brv_cconv_rows(const float* in, const float* in2, int64_t in_seg, const void* wp, const float* bias, float* out,
                   float* out2, int64_t out_seg, int64_t B, int64_t C, int64_t M, int64_t Hin, int64_t Win,
                   int32_t transposed, brv_stream_t stream);
/* Weight gradient of both forms of brv_cconv_rows: out[a][10 c + 2 i + j] = sum_{b,h,w} small[b][a][h][w] *
 * big[b][c][2h - 2 + i][w + j], small (B, A, Hs, Ws), big (B, C, 2 Hs, Ws + 1), out (A, 10 C) fp32 (overwritten).
 * Conv2d: small = dy, big = x; ConvTranspose2d: small = x, big = dy (brever/models/dccrn/dccrn.py:225-235
 * under autograd). seg > 0: `small` is the channel concatenation [small[:, :seg], small2[:, :seg], small[:, seg:],
 * small2[:, seg:]] of two (B, 2 seg, Hs, Ws) tensors (the decoder's skip concatenation, dccrn.py:213-217),
 * A = 4 seg; seg = 0: small2 is ignored. bf16 operands, fp32 sums: the (b, h) range is split over workgroups whose
 * partial matrices (workspace of brv_cconv_wgrad_workspace_bytes bytes) a second launch adds in split order. */
int64_t brv_cconv_wgrad_workspace_bytes(int64_t B, int64_t A, int64_t C, int64_t Hs);
int brv_cconv_wgrad(const float* small, const float* small2, const float* big, float* out, void* workspace,
                    int64_t B, int64_t A, int64_t C, int64_t Hs, int64_t Ws, int64_t seg, brv_stream_t stream);
/* The same two products with the IMAGES stored as bf16 (in / in2; small / small2 / big -- all alike): the values the
 * fp32 forms round their operands to on the way into LDS. Results are bit-identical to the fp32 forms given images
 * that hold bf16-representable values. Producers: brv_batchnorm2d_forward_bf16 / brv_batchnorm2d_backward_bf16
 * below; reference: what torch.autocast holds between the layers of brever/models/dccrn/dccrn.py:238-292. Outputs
 * stay fp32. bf16 images need no conversion on the way into LDS: brv_cconv_rows_bf16 stages them by LDS-DMA
 * (csrc/cconv_dma.cuh) wherever C is a multiple of 8, and that path fetches whole 16-byte pieces around a row's ends
 * through range-checked descriptors that reach 16 bytes in front of and behind the tensor.
 * brv_cconv_wgrad_bf16 does the same (csrc/cconv_wgrad_dma.cuh) unless its two-source segments are not whole
 * groups of 8 channels.
 * CONTRACT: every bf16 image handed to these two functions (in, in2; small, small2, big) must have 16 READABLE bytes
 * on both sides (allocate 8 elements more at each end; their contents do not matter). */
int brv_cconv_rows_bf16(const void* in, const void* in2, int64_t in_seg, const void* wp, const float* bias, float* out,
                        float* out2, int64_t out_seg, int64_t B, int64_t C, int64_t M, int64_t Hin, int64_t Win,
                        int32_t transposed, brv_stream_t stream);
int brv_cconv_wgrad_bf16(const void* small, const void* small2, const void* big, float* out, void* workspace,
                         int64_t B, int64_t A, int64_t C, int64_t Hs, int64_t Ws, int64_t seg, brv_stream_t stream);
/* brv_cconv_rows / brv_cconv_rows_bf16 with the element types as arguments: in_bf16 selects the image type (0: fp32,
 * 1: bf16 -- with the readable-slack contract above), out_bf16 the type out / out2 are written in (1: rounded to bf16:
 * what torch.autocast makes of a convolution output; read by brv_batchnorm2d_forward_bf16io / _backward_bf16io). */
int brv_cconv_rows_ex(const void* in, const void* in2, int64_t in_seg, const void* wp, const float* bias, void* out,
                      void* out2, int64_t out_seg, int64_t B, int64_t C, int64_t M, int64_t Hin, int64_t Win,
                      int32_t transposed, int32_t in_bf16, int32_t out_bf16, brv_stream_t stream);
/* brv_gemm_bf16 with bf16 tensors in memory: flags bit 0: b holds bf16 elements, bit 2: a does, bit 1:
 * d is written as bf16 (no accumulate, no split reduction); strides count elements. Used with
 * brv_im2col_bf16 / brv_col2im_bf16 (same arguments as brv_im2col / brv_col2im, the column matrix
 * in bf16) by the use_amp path of DCCRN's convolutions: the column matrix is the largest tensor
 * of a convolution-as-product. */
int brv_gemm_bf16_mixed(const void* a, const void* b, void* d, int64_t batch, int64_t M, int64_t N,
                        int64_t K, int64_t lda, int64_t ldb, int64_t ldd, int64_t a_batch_stride,
                        int64_t b_batch_stride, int64_t d_batch_stride, int trans_a, int trans_b,
                        int64_t kbatch, int64_t a_kbatch_stride, int64_t b_kbatch_stride,
                        const float* row_bias, int accumulate, int flags, brv_stream_t stream);
int brv_im2col_bf16(const float* x, void* col, int64_t B, int64_t C, int64_t H, int64_t W, int64_t kh,
                    int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t Ho,
                    int64_t Wo, brv_stream_t stream);
int brv_col2im_bf16(const void* col, const float* bias, float* y, int64_t B, int64_t C, int64_t H,
                    int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                    int64_t Ho, int64_t Wo, brv_stream_t stream);

/* ---- FFNN mask model and log-mel features (models/ffnn/ffnn.py:72-203,
 * modules/features.py:142-205); fp32, (B, rows, frames) / complex64 (B, channels, bins*frames)
 * brv_fbe_power: mean over channels of |spec|^2; brv_compress: mode 1 log(x+eps), 2 cube root, 3 square root;
 * brv_irm: (1 + bg/(fg+eps))^-1/2; brv_stack_frames: delayed copies with first-frame fill;
 * brv_static_norm / brv_cumulative_norm: StaticNormalizer / CumulativeNormalizer;
 * brv_relu_dropout_*: ReLU followed by dropout with a caller-supplied keep mask (null: none)
 * scaled by `scale`; brv_sigmoid_*; brv_row_sum: bias gradient sum over batch and frames;
 * brv_masked_mean_spec: mask * mean over channels of a complex spectrum (FFNN._enhance). */
int brv_fbe_power(const float* spec, float* out, int64_t B, int64_t C, int64_t n, brv_stream_t stream);
int brv_compress(const float* x, float* out, int64_t n, int mode, float eps, brv_stream_t stream);
/* The other members of the FeatureExtractor family (modules/features.py:142-262): binaural:
 * ILD (mode 0) / IPD (mode 1) of a (B, 2, n) complex64 spectrum -> (B, n); col_normalize: the
 * 'pdf' normalisation over the filter axis of (B, M, T), in place; deltas: (B, M, T) ->
 * (B, 3M, T) = [x | first | second difference along the frames, zero left padding] (the
 * delta / double-delta rows of the mfcc features; the DCT itself is a brv_matmul_f32). */
/* Interaural coherence (FeatureExtractor.ic, features.py:263-293) of spec (B, 2, bins, F) complex
 * -> (B, bins, F): first-order recursive smoothing of the auto- / cross-power spectra along the
 * frames with coefficient alpha (torchaudio lfilter semantics incl. its output clamp to [-1, 1]),
 * then |phi_lr|^2 / (phi_ll phi_rr); brv_compress mode 3 is the final square root. */
int brv_interaural_coherence(const float* spec, float* out, int64_t B, int64_t bins, int64_t F,
                             float alpha, brv_stream_t stream);
int brv_binaural(const float* spec, float* out, int64_t B, int64_t n, int mode, float eps,
                 brv_stream_t stream);
int brv_col_normalize(float* x, int64_t B, int64_t M, int64_t T, float eps, brv_stream_t stream);
int brv_deltas(const float* x, float* out, int64_t B, int64_t M, int64_t T, brv_stream_t stream);
int brv_irm(const float* fg, const float* bg, float* out, int64_t n, float eps, brv_stream_t stream);
int brv_stack_frames(const float* x, float* out, int64_t B, int64_t nf, int64_t T, int64_t stacks,
                     brv_stream_t stream);
int brv_static_norm(const float* x, const float* mean, const float* stdv, float* out, int64_t B,
                    int64_t rows, int64_t T, brv_stream_t stream);
int brv_cumulative_norm(const float* x, float* out, int64_t nrows, int64_t T, float eps,
                        brv_stream_t stream);
int brv_relu_dropout_forward(const float* x, const float* mask, float* out, int64_t n, float scale,
                             brv_stream_t stream);
int brv_relu_dropout_backward(const float* x, const float* mask, const float* dy, float* dx,
                              int64_t n, float scale, brv_stream_t stream);
/* nn.Dropout with a caller-drawn keep mask (UNetBlock.dropout, sgmse/net.py:409; forward on x,
 * backward on dy): out = x*mask*scale. */
int brv_dropout_apply(const float* x, const float* mask, float* out, int64_t n, float scale,
                      brv_stream_t stream);
int brv_sigmoid_forward(const float* x, float* out, int64_t n, brv_stream_t stream);
int brv_sigmoid_backward(const float* y, const float* dy, float* dx, int64_t n, brv_stream_t stream);
int brv_row_sum(const float* x, float* out, int64_t B, int64_t M, int64_t T, brv_stream_t stream);
int brv_masked_mean_spec(const float* spec, const float* mask, float* out, int64_t B, int64_t C,
                         int64_t n, brv_stream_t stream);

/* ---- DCCRN building blocks, forward values (models/dccrn/dccrn.py:28-358) --------------
 * NCHW fp32 as in the reference; batch strides let real / imaginary halves of a wider
 * tensor be addressed in place. conv / conv_transpose: y (+)= sign*(op(x, w) + bias)
 * (accumulate != 0 adds to y): the four real convolutions of a ComplexWrapper are two
 * such calls per output half. batchnorm2d: training != 0 uses batch statistics and updates
 * the running estimates (momentum), else the running ones; an optional scalar PReLU
 * follows. lstm_recurrent: gates_in (B, T, 4H) = W_ih x for all steps, bias (4H) =
 * b_ih + b_hh, y (B, T, H), zero initial state, torch gate order; `groups` independent LSTMs
 * in one launch: the B items are `groups` consecutive sets of B/groups items, w_hh
 * (groups, 4H, H) and bias (groups, 4H) hold one parameter set per group; act / cs (nullable) receive the gate
 * activations and cell states for the backward pass. dccrn_apply_mask: DCCRN.apply_mask. */
int brv_conv2d_forward(const float* x, const float* w, const float* bias, float* y, int64_t B,
                       int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t kh, int64_t kw,
                       int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t x_batch_stride,
                       int64_t y_batch_stride, int accumulate, float sign, brv_stream_t stream);
int brv_conv_transpose2d_forward(const float* x, const float* w, const float* bias, float* y,
                                 int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                                 int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph,
                                 int64_t pw, int64_t oph, int64_t opw, int64_t x_batch_stride,
                                 int64_t y_batch_stride, int accumulate, float sign,
                                 brv_stream_t stream);
int brv_batchnorm2d_forward(const float* x, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, const float* prelu_slope,
                            float* y, float* save_mean, float* save_invstd, int64_t B, int64_t C,
                            int64_t HW, float eps, float momentum, int training,
                            brv_stream_t stream);
int brv_lstm_recurrent_forward(const float* gates_in, const float* w_hh, const float* bias,
                               float* y, float* act, float* cs, int64_t B, int64_t T, int64_t H,
                               int64_t groups, brv_stream_t stream);
/* Backward pieces (autograd of the above). conv2d_wgrad: dw (Cout, Cin, kh, kw) and dbias
 * from x and dy (ConvTranspose2d: pass its output gradient as x and its input as dy; data
 * gradients are the forward kernels of the opposite operation with the same weights).
 * batchnorm2d_backward: training-mode batch norm followed by the optional PReLU;
 * dslope_partial (C) holds per-channel partial sums of the slope gradient.
 * lstm_recurrent_backward: act (B, T, 4H) gate activations and cs (B, T, H) cell states
 * saved by the forward (act / cs non-null there), dy (B, T, H) -> dgates (B, T, 4H).
 * dccrn_apply_mask_backward: gradient wrt the mask. istft_env_divide: dy / window-square
 * envelope of torch.istft (first step of the adjoint of brv_istft_backward). */
int brv_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, int64_t B,
                     int64_t Cin, int64_t H, int64_t W, int64_t Cout, int64_t Ho, int64_t Wo,
                     int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                     int64_t x_batch_stride, int64_t dy_batch_stride, int accumulate, float sign,
                     brv_stream_t stream);
int brv_batchnorm2d_backward(const float* x, const float* dy, const float* save_mean,
                             const float* save_invstd, const float* gamma, const float* beta,
                             const float* prelu_slope, float* dx, float* dgamma, float* dbeta,
                             float* dslope_partial, int64_t B, int64_t C, int64_t HW,
                             brv_stream_t stream);
/* The two batch-norm passes with their element-wise OUTPUT written as bf16 (use_amp: the only readers are the row
 * convolutions above, which round to bf16 anyway -- nn.BatchNorm2d + nn.PReLU of EncoderBlock / DecoderBlock,
 * dccrn.py:238-292, under torch.autocast). HW must be a multiple of 4 (-1 otherwise). Statistics, parameter gradients
 * and all arithmetic are the fp32 forms'. backward: dx_sums (C) or NULL: per-channel sums of the UNROUNDED dx (the
 * bias gradient of the convolution in front of the norm; replaces a brv_row_sum pass over dx). */
int brv_batchnorm2d_forward_bf16(const float* x, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, const float* prelu_slope,
                                 void* y16, float* save_mean, float* save_invstd, int64_t B, int64_t C,
                                 int64_t HW, float eps, float momentum, int training,
                                 brv_stream_t stream);
int brv_batchnorm2d_backward_bf16(const float* x, const float* dy, const float* save_mean,
                                  const float* save_invstd, const float* gamma, const float* beta,
                                  const float* prelu_slope, void* dx16, float* dgamma, float* dbeta,
                                  float* dslope_partial, float* dx_sums, int64_t B, int64_t C, int64_t HW,
                                  brv_stream_t stream);
/* ... and with the norm's INPUT x stored as bf16 too (the convolution output as brv_cconv_rows_ex(out_bf16 = 1) writes
 * it): the same arithmetic on the widened values. */
int brv_batchnorm2d_forward_bf16io(const void* x16, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, const float* prelu_slope,
                                   void* y16, float* save_mean, float* save_invstd, int64_t B, int64_t C,
                                   int64_t HW, float eps, float momentum, int training,
                                   brv_stream_t stream);
int brv_batchnorm2d_backward_bf16io(const void* x16, const float* dy, const float* save_mean,
                                    const float* save_invstd, const float* gamma, const float* beta,
                                    const float* prelu_slope, void* dx16, float* dgamma, float* dbeta,
                                    float* dslope_partial, float* dx_sums, int64_t B, int64_t C, int64_t HW,
                                    brv_stream_t stream);
/* The backward pass with the element types as arguments and an optional SECOND gradient dy2 with respect to the output,
 * added on the fly (an encoder block's output feeds the next block and the decoder's skip input, dccrn.py:205-217: no
 * pass that sums the two gradients); dy_bf16: dy / dy2 hold bf16 elements (the gradient of a bf16 activation as
 * torch.autocast hands it over). HW a multiple of 4 whenever dy2, dx_sums or a bf16 type is used. */
int brv_batchnorm2d_backward_ex(const void* x, int32_t x_bf16, const void* dy, const void* dy2, int32_t dy_bf16,
                                const float* save_mean,
                                const float* save_invstd, const float* gamma, const float* beta,
                                const float* prelu_slope, void* dx, int32_t dx_bf16, float* dgamma, float* dbeta,
                                float* dslope_partial, float* dx_sums, int64_t B, int64_t C, int64_t HW,
                                brv_stream_t stream);
int brv_lstm_recurrent_backward(const float* act, const float* cs, const float* w_hh,
                                const float* dy, float* dgates, int64_t B, int64_t T, int64_t H,
                                int64_t groups, brv_stream_t stream);
/* The same recurrences with the step's matrix-vector product on the bf16 MFMA (use_amp: W_hh and the hidden state /
 * the gate gradients are rounded to bf16 as operands, fp32 accumulation; gate math, cell state and all tensors fp32 --
 * reference: nn.LSTM under torch.autocast, brever/models/dccrn/dccrn.py:293-311 with training.use_amp). Same arguments
 * and layouts; H must satisfy brv_lstm_recurrent_bf16_supported (128). */
int brv_lstm_recurrent_bf16_supported(int64_t H);
int brv_lstm_recurrent_forward_bf16(const float* gates_in, const float* w_hh, const float* bias,
                                    float* y, float* act, float* cs, int64_t B, int64_t T, int64_t H,
                                    int64_t groups, brv_stream_t stream);
int brv_lstm_recurrent_backward_bf16(const float* act, const float* cs, const float* w_hh,
                                     const float* dy, float* dgates, int64_t B, int64_t T, int64_t H,
                                     int64_t groups, brv_stream_t stream);
int brv_dccrn_apply_mask_backward(const float* xr, const float* xi, const float* mr,
                                  const float* mi, const float* gout, float* dmr, float* dmi,
                                  int64_t n, brv_stream_t stream);
/* the same two maps for a whole batch in one launch: x, mask, dmask (B, 2 n) with the real plane first, out / gout
 * (B, n, 2) */
int brv_dccrn_apply_mask_batched(const float* x, const float* mask, float* out, int64_t B, int64_t n,
                                 brv_stream_t stream);
int brv_dccrn_apply_mask_backward_batched(const float* x, const float* mask, const float* gout, float* dmask,
                                          int64_t B, int64_t n, brv_stream_t stream);
int brv_istft_env_divide(const float* dy, const float* window, float* out, int64_t rows,
                         int64_t length, int64_t frame_length, int64_t hop_length, int64_t frames,
                         brv_stream_t stream);
/* The complex combination of a ComplexLSTM layer (brever/models/dccrn/dccrn.py:330-358): o (2, 2, n) = module m on input
 * half h; real = o[0][0] - o[1][1], imag = o[0][1] + o[1][0]; backward: dout from the two gradients. n a multiple of 4,
 * 16-byte aligned pointers. */
int brv_complex_mix_forward(const float* o, float* real, float* imag, int64_t n, brv_stream_t stream);
int brv_complex_mix_backward(const float* greal, const float* gimag, float* dout, int64_t n, brv_stream_t stream);
int brv_combine(const float* a, const float* b, float* out, int64_t n, float sign,
                brv_stream_t stream);
int brv_dccrn_apply_mask(const float* xr, const float* xi, const float* mr, const float* mi,
                         float* out, int64_t n, brv_stream_t stream);

/* Convolution as matrix products (ComplexWrapper(nn.Conv2d / nn.ConvTranspose2d),
 * models/dccrn/dccrn.py:221-231): im2col writes col (B, C*kh*kw, Ho*Wo) from x (B, C, H, W);
 * col2im is its adjoint (+ per-channel bias) onto y (B, C, H, W) from a column grid Ho x Wo;
 * the four real convolutions of a complex one are ONE brv_gemm_f32 with the weight
 * wc (2R x 2C) = [[wr, -sign*wi], [sign*wi, wr]] built by complex_weight_pack from wr, wi
 * (R x C each); complex_weight_unpack is the adjoint (dwr, dwi from dwc). */
int brv_im2col(const float* x, float* col, int64_t B, int64_t C, int64_t H, int64_t W, int64_t kh,
               int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t Ho, int64_t Wo,
               brv_stream_t stream);
int brv_col2im(const float* col, const float* bias, float* y, int64_t B, int64_t C, int64_t H,
               int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
               int64_t Ho, int64_t Wo, brv_stream_t stream);
int brv_complex_weight_pack(const float* wr, const float* wi, float* wc, int64_t R, int64_t C,
                            float sign, brv_stream_t stream);
int brv_complex_weight_unpack(const float* dwc, float* dwr, float* dwi, int64_t R, int64_t C,
                              float sign, brv_stream_t stream);
/* the bias of that packed layer, out (2C) = [br - bi | br + bi] (each module of ComplexWrapper adds its own bias,
 * dccrn.py:231-235), and its adjoint on the channel sums [s_r | s_i] of the output gradient: d br = s_r + s_i,
 * d bi = s_i - s_r */
int brv_complex_bias_pack(const float* br, const float* bi, float* out, int64_t C, brv_stream_t stream);
int brv_complex_bias_unpack(const float* sums, float* dbr, float* dbi, int64_t C, brv_stream_t stream);

/* ComplexBatchNorm2d (models/dccrn/complex_batchnorm.py:29-215) on x (B, 2C, HW), real half
 * first: cplx_moments writes the five per-channel means (5, C) = E[xr], E[xi], E[xr^2],
 * E[xi^2], E[xr xi]; the 2x2 whitening / affine scalars derived from them are applied by
 * cplx_affine_forward: y_r = A[0]xr + A[1]xi + o[0], y_i = A[2]xr + A[3]xi + o[1] (A (4, C),
 * o (2, C)) followed by the optional scalar PReLU; cplx_affine_backward returns dx, dA, d_o and
 * per-channel partial sums of the slope gradient; cplx_moments_backward turns the gradient with
 * respect to the five means (already divided by B*HW) into dx. */
int brv_cplx_moments(const float* x, float* moments, int64_t B, int64_t C, int64_t HW,
                     brv_stream_t stream);
int brv_cplx_affine_forward(const float* x, const float* A, const float* o, const float* prelu_slope,
                            float* y, int64_t B, int64_t C, int64_t HW, brv_stream_t stream);
int brv_cplx_affine_backward(const float* x, const float* dy, const float* A, const float* o,
                             const float* prelu_slope, float* dx, float* dA, float* d_o,
                             float* dslope_partial, int64_t B, int64_t C, int64_t HW,
                             brv_stream_t stream);
int brv_cplx_moments_backward(const float* x, const float* gm, float* dx, int64_t B, int64_t C,
                              int64_t HW, brv_stream_t stream);

/* Causal (cumulative) group normalisation, fp32 (modules/normalization.py:5-62: CausalGroupNorm,
 * CausalLayerNorm = 1 group, CausalInstanceNorm = one group per channel): x (B, C, inner, T)
 * with the frames last, every frame normalised with the statistics of its group over all frames
 * up to it. stats (B*groups, T, 2) receives (mean, rstd) for the backward pass; scratch:
 * brv_causal_groupnorm_scratch_bytes(); uv_scratch: B*groups*T*2 floats. */
int64_t brv_causal_groupnorm_scratch_bytes(int64_t B, int64_t groups, int64_t T);
int brv_causal_groupnorm_forward(const float* x, const float* gain, const float* bias, float* y,
                                 float* stats, void* scratch, int64_t B, int64_t C, int64_t inner,
                                 int64_t T, int64_t groups, float eps, brv_stream_t stream);
int brv_causal_groupnorm_backward(const float* x, const float* dy, const float* gain,
                                  const float* stats, float* dx, float* dgain, float* dbias,
                                  void* scratch, float* uv_scratch, int64_t B, int64_t C,
                                  int64_t inner, int64_t T, int64_t groups, brv_stream_t stream);

/* ---- LSTM recurrence for many short chains (nn.LSTM as used by tfgridnet.py:200-216): 16
 * chains per workgroup, one exact-fp32 MFMA product per step, hidden size 128 only
 * (brv_lstm_tile_supported). Same arguments as brv_lstm_recurrent_forward / _backward EXCEPT the
 * gate layout of gates_in, act and dgates, which is interleaved (column = 4*unit + gate) instead
 * of torch's gate-major order; w_hh and bias stay in torch's layout. Bit g of reverse_mask makes
 * group g run over the frames backwards (the second direction of a bidirectional nn.LSTM, no
 * flipped copies). y / dy element (group g, chain c of the group, frame t, unit u) lives at
 * g*group_offset + (c*T + t)*ld + u: ld = H, group_offset = chains_per_group*T*H is the plain
 * (groups, chains, T, H) layout; ld = 2H, group_offset = H writes both directions straight into
 * nn.LSTM's (chains, T, 2H) output. lowp != 0 (use_amp): W_hh and the recurrent operand (h or the
 * gate gradients) are rounded to bf16 for the MFMA; accumulation, cell state, gate arithmetic and
 * every tensor in memory stay fp32; lowp == 2 additionally keeps gates_in, act and dgates as bf16 tensors
 * (the two largest streams of the then HBM-bound kernel; the input projection writes bf16 through
 * brv_gemm_bf16_mixed). */
int brv_lstm_tile_supported(int64_t H);
int brv_lstm_tile_forward(const float* gates_in, const float* w_hh, const float* bias, float* y,
                          float* act, float* cs, int64_t B, int64_t T, int64_t H, int64_t groups,
                          int64_t reverse_mask, int64_t y_ld, int64_t y_group_offset, int lowp,
                          brv_stream_t stream);
int brv_lstm_tile_backward(const float* act, const float* cs, const float* w_hh, const float* dy,
                           float* dgates, int64_t B, int64_t T, int64_t H, int64_t groups,
                           int64_t reverse_mask, int64_t dy_ld, int64_t dy_group_offset, int lowp,
                           brv_stream_t stream);

/* ---- TF-GridNet row operators (models/tfgridnet/tfgridnet.py). rownorm: layer normalisation of
 * `rows` contiguous rows of n floats, optionally behind a PReLU, with gain / bias (groups, n); the
 * group (and PReLU slope, nullable = no PReLU) of row r is (r / inner) % groups. Replaces
 * nn.LayerNorm(emb_dim) (tfgridnet.py:199,213), LayerNormalization4DCF (tfgridnet.py:356-380) and
 * AllHeadPReLULayerNormalization4DCF (tfgridnet.py:383-415) on rows laid out so that the
 * normalised axes are contiguous. stats (rows, 2) = (mean, rstd). backward: dx, dgain / dbias
 * (groups, n), dslope_rows (rows) = per-row PReLU slope gradient (written only with a slope);
 * scratch: brv_rownorm_scratch_bytes(). row_std: unbiased standard deviation of each row
 * (torch.std, tfgridnet.py:108); row_scale: y = x * s[row] or x / s[row] (tfgridnet.py:109,128). */
int brv_rownorm_forward(const float* x, const float* slope, const float* gain, const float* bias,
                        float* y, float* stats, int64_t rows, int64_t n, int64_t inner,
                        int64_t groups, float eps, brv_stream_t stream);
int64_t brv_rownorm_scratch_bytes(int64_t n, int64_t groups);
int brv_rownorm_backward(const float* x, const float* dy, const float* slope, const float* gain,
                         const float* stats, float* dx, float* dgain, float* dbias,
                         float* dslope_rows, void* scratch, int64_t rows, int64_t n, int64_t inner,
                         int64_t groups, brv_stream_t stream);
/* out (batch, cols) = column sums of x (batch, rows, cols), fp32, fixed summation order (the bias
 * gradients of nn.Linear / nn.LSTM on row-major activations); scratch: brv_col_sum_scratch_bytes(). */
int64_t brv_col_sum_scratch_bytes(int64_t batch, int64_t cols);
/* y (M x N) = x (M x K) @ op(w) + bias[n] (bias may be NULL; accumulate: y += instead) for the narrow linear
 * layers of the TF-GridNet grid blocks and their data gradients (reference tfgridnet.py GridNetBlock intra / inter
 * linear: M = batch x frames x bands, K, N in 16 .. 64): one thread per row, fp32 FMAs, weights through the scalar
 * cache. op(w)[k][n] = w[k*ldw + n], or w[n*ldw + k] with trans_b (nn.Linear's (out, in) weight). Supported:
 * N in {16, 32, 64}, K a multiple of 4 up to 64, M >= 4096, 16-byte aligned x / y with lda, ldd multiples of 4
 * (brv_linear_small_supported tells the shape part); -1 otherwise. */
int brv_linear_small_supported(int64_t M, int64_t N, int64_t K);
/* The weight gradient of those layers: d (MI x NJ, row stride ldd) = sum over rows r of a[r][i] * b[r][j]
 * (a rows x MI, b rows x NJ, row-major with strides lda / ldb: a = dy, b = x for nn.Linear's (out, in) weight).
 * MI, NJ in {16, 32}, rows >= 4096, 16-byte aligned operands with strides multiples of 4; the rows are cut into
 * slices whose partial matrices (scratch: brv_linear_small_wgrad_scratch_bytes) are added in slice order --
 * bitwise repeatable. -1 for anything else. */
int brv_linear_small_wgrad_supported(int64_t rows, int64_t MI, int64_t NJ);
int64_t brv_linear_small_wgrad_scratch_bytes(int64_t MI, int64_t NJ);
int brv_linear_small_wgrad(const float* a, const float* b, float* d, void* scratch, int64_t rows, int64_t MI,
                           int64_t NJ, int64_t lda, int64_t ldb, int64_t ldd, brv_stream_t stream);
int brv_linear_small(const float* x, const float* w, const float* bias, float* y, int64_t M, int64_t N, int64_t K,
                     int64_t lda, int64_t ldw, int64_t ldd, int trans_b, int accumulate, brv_stream_t stream);
int brv_col_sum(const float* x, float* out, void* scratch, int64_t batch, int64_t rows, int64_t cols,
                brv_stream_t stream);
/* The same sums of a bf16 (rows x cols) matrix (cols a multiple of 8, 16-byte aligned; the gate gradients the use_amp
 * recurrences keep in bf16), fp32 accumulation and output, same scratch; -1 for other shapes. */
int brv_col_sum_bf16(const void* x, float* out, void* scratch, int64_t batch, int64_t rows, int64_t cols,
                     brv_stream_t stream);
int brv_row_std(const float* x, float* out, int64_t rows, int64_t n, brv_stream_t stream);
/* Head split / merge of TF-GridNet's attention (reference brever/models/tfgridnet/tfgridnet.py:315-353; there a
 * view + permute on (B, C, T, F) tensors): merge = 0: in (B, T, F, H, E) channels-last -> out (B, H, T, E, F), the
 * (items, frames, features) rows of the head norms and attention products; merge = 1: the inverse. One launch, both
 * sides in contiguous runs; H E (F | 1) floats of LDS (brv_head_permute_supported). */
int brv_head_permute_supported(int64_t F, int64_t H, int64_t E);
int brv_head_permute(const float* in, float* out, int64_t B, int64_t T, int64_t F, int64_t H, int64_t E, int merge,
                     brv_stream_t stream);
int brv_row_scale(const float* x, const float* s, float* y, int64_t rows, int64_t n, int divide,
                  brv_stream_t stream);

/* ---- SGMSE+ score network building blocks, forward values (models/sgmse/net.py:12-477,
 * modules/resampling.py:8-61). groupnorm_fold: nn.GroupNorm on x + add_bc[b][c] (nullable; the
 * noise-embedding term of UNetBlock) reduced to a per-(item, channel) affine scale / shift
 * (B, C), optionally followed by the ADM modulation (1 + adm_scale)*norm + adm_shift
 * (net.py:405-407); scratch: brv_groupnorm_scratch_bytes(), ZERO on entry and left zero on
 * exit (clear it once after allocation, then reuse it call after call on one stream).
 * affine_act applies it
 * (y = act(scale*x + shift)); brv_conv2d_mfma_forward can apply it on load instead.
 * softmax_rows: attention weights;
 * fir_resample2d: Resample.forward on `planes` = B*C images (up: transposed, kernel*gain);
 * axpby: alpha*a + beta*b; fourier_features: GaussianFourierProjection. */
int64_t brv_groupnorm_scratch_bytes(int64_t B, int64_t groups);
int brv_groupnorm_fold(const float* x, const float* add_bc, const float* gamma, const float* beta,
                       const float* adm_scale, const float* adm_shift, void* scratch, float* scale,
                       float* shift, float* mu_bc, float* rstd_bc, int64_t B, int64_t C, int64_t HW,
                       int64_t groups, float eps, brv_stream_t stream);
/* Backward of y = act(GroupNorm(x + add)) (autograd of net.py:395-412 for SGMSE+ training):
 * mu_bc / rstd_bc (nullable in the forward call) are the per-(item, channel) centre (group
 * mean - add) and inverse deviation saved by brv_groupnorm_fold. Writes dx, the per-(item,
 * channel) sums s1 = sum dpre and s2 = sum dpre*xhat (d beta and d gamma are their sums over
 * the batch), d add (nullable); coef_scratch: 3*B*C floats. silu / softmax_rows backward:
 * autograd of F.silu and of the attention softmax. */
int brv_groupnorm_backward(const float* x, const float* dy, const float* scale_bc,
                           const float* shift_bc, const float* mu_bc, const float* rstd_bc,
                           const float* gamma, float* dx, float* s1_bc, float* s2_bc, float* dadd_bc,
                           float* coef_scratch, int64_t B, int64_t C, int64_t HW, int64_t groups,
                           int act_silu, brv_stream_t stream);
/* Backward of brv_affine_act (the ADM modulation of UNetBlock, net.py:405-407): dx and the
 * per-(item, channel) gradients of scale and shift; zeros_bc / ones_bc: (B, C) constants. */
int brv_affine_act_backward(const float* x, const float* dy, const float* scale_bc,
                            const float* shift_bc, const float* zeros_bc, const float* ones_bc,
                            float* dx, float* dscale_bc, float* dshift_bc, int64_t B, int64_t C,
                            int64_t HW, int act_silu, brv_stream_t stream);
int brv_silu_backward(const float* x, const float* dy, float* dx, int64_t n, brv_stream_t stream);
int brv_softmax_rows_backward(const float* p, const float* dy, float* dx, int64_t rows, int64_t cols,
                              brv_stream_t stream);
int brv_affine_act(const float* x, const float* scale_bc, const float* shift_bc, float* y,
                   int64_t B, int64_t C, int64_t HW, int act_silu, brv_stream_t stream);
int brv_silu(const float* x, float* y, int64_t n, brv_stream_t stream);
int brv_softmax_rows(const float* x, float* y, int64_t rows, int64_t cols, brv_stream_t stream);
int brv_fir_resample2d(const float* x, const float* kernel, float* y, int64_t planes, int64_t H,
                       int64_t W, int64_t Ho, int64_t Wo, int64_t K, int64_t pad_h, int64_t pad_w,
                       int up, float gain, brv_stream_t stream);
int brv_axpby(const float* a, float alpha, const float* b, float beta, float* out, int64_t n,
              brv_stream_t stream);
int brv_fourier_features(const float* x, const float* b, float* out, int64_t n, int64_t m,
                         brv_stream_t stream);

/* Stride-1 "same" convolution (ksize 1 or 3) on the fp16 MFMA with fp32 accumulation -- the
 * precision class of the reference's fp16 autocast inference (models/sgmse/sgmse.py:190-193)
 * -- for the convolutions of UNetBlock / AttentionBlock (net.py:352-452). x / y fp32 NCHW
 * with batch strides; wp: weights (Cout, Cin, k, k) packed once by brv_conv2d_pack_f16 into
 * brv_conv2d_packed_size() halves. Fused on the way in (nullable): in_scale[b][ci]*x +
 * in_shift[b][ci] then SiLU if in_silu = a folded GroupNorm, padding stays zero; on the way out:
 * y = out_scale*(conv + bias + res), res (nullable) laid out like y. */
int64_t brv_conv2d_packed_size(int64_t Cout, int64_t Cin, int64_t ksize);
int brv_conv2d_pack_f16(const float* w, void* wp, int64_t Cout, int64_t Cin, int64_t ksize,
                        brv_stream_t stream);
int brv_conv2d_mfma_forward(const float* x, const void* wp, const float* bias, const float* res,
                            const float* in_scale, const float* in_shift, int in_silu, float* y,
                            int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t Cout,
                            int64_t ksize, int64_t x_batch_stride, int64_t y_batch_stride,
                            float out_scale, brv_stream_t stream);

/* ---- SGMSE+ score network under use_amp, channels-last fp16 activations ------------------------
 * The reference runs the score network under fp16 autocast (models/sgmse/sgmse.py:190-193): the
 * tensors between its convolutions are fp16 there too. Here they are (B, H, W, Cs) fp16, Cs a
 * multiple of 8, channels >= C zero; the 4-channel progressive branch and the network's input /
 * output stay (B, C, H, W) fp32.
 * brv_conv_nhwc_forward: 3x3 stride-1 "same" convolution (UNetBlock.conv_1 / conv_2, net.py:352-422)
 *   on the fp16 MFMA, fp32 accumulation: y = out_scale*(conv(act([x1 | x2])) + bias + res), act =
 *   silu?(in_scale[b][ci]*x + in_shift[b][ci]) when in_scale is given (a GroupNorm folded by
 *   brv_groupnorm_fold_chan; needs Cin % 32 == 0), x2 (nullable) = a second tensor concatenated
 *   along the channels (the U-Net's skip connections, net.py:330-335; needs C1 % 32 == 0);
 *   wp from brv_conv_nhwc_pack. Cout % 4 == 0, channel strides % 8 == 0. stats (nullable, (B, Cout,
 *   2) fp64, cleared by the caller) += per-channel (sum, sum of squares) of y: the statistics of
 *   the next GroupNorm come out of the producing convolution's epilogue.
 * brv_nhwc_conv1x1_*: UNetBlock.skip_conv on [x1 | x2].
 * brv_nhwc_chan_stats: sums[b][c_off + c][0..1] += per-channel (sum, sum of squares) over the
 *   pixels (fp64; clear `sums` (B, Ctot, 2) first); brv_groupnorm_fold_chan: GroupNorm of x +
 *   add_bc reduced to scale / shift (B, C) from those sums (same arithmetic as brv_groupnorm_fold).
 * brv_nhwc_affine_act / _fir_resample2d / _axpby: the channels-last forms of brv_affine_act,
 *   brv_fir_resample2d, brv_axpby. brv_nhwc_conv3x3_small: 3x3 convolution to <= 8 channels,
 *   (B, Cout, H, W) fp32 out = [y_in +] conv(act(x)) + bias (AuxiliaryUp.conv, the output
 *   convolution; net.py:455-477); w16 = 9*Cout*C halves from brv_nhwc_conv3x3_small_pack, C % 8 == 0. brv_nhwc_add_pointwise: y = out_scale*(x + bias + W aux), aux
 *   (B, K <= 8, HW) fp32 (AuxiliaryDown, net.py:425-452). */
int64_t brv_conv_nhwc_packed_size(int64_t Cout, int64_t Cin, int64_t ksize);
int brv_conv_nhwc_pack(const float* w, void* wp, int64_t Cout, int64_t Cin, int64_t ksize,
                       brv_stream_t stream);
int brv_conv_nhwc_forward(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                          int64_t C2s, const void* wp, const float* bias, const void* res,
                          int64_t Crs, const float* in_scale, const float* in_shift, int in_silu,
                          void* y, int64_t Cys, int64_t B, int64_t H, int64_t W, int64_t Cout,
                          int64_t ksize, float out_scale, double* stats, brv_stream_t stream);
/* The same convolution with the GroupNorm that feeds it given by its ingredients -- per-channel sums
 * of the one or two sources (brv_nhwc_chan_stats / the producing convolution's `stats`), the
 * embedding term add_bc (B, Cin), gamma / beta, the ADM modulation -- instead of a folded scale /
 * shift: when every workgroup's tiles belong to one item the kernel folds them itself (no launch
 * between producer and consumer), else brv_groupnorm_fold_chan2 runs into fold_ws (2*B*Cin floats). */
int brv_conv_nhwc_forward_gn(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                             int64_t C2s, const void* wp, const float* bias, const void* res,
                             int64_t Crs, const double* sums1, const double* sums2,
                             const float* add_bc, const float* gamma, const float* beta,
                             const float* adm_scale, const float* adm_shift, int64_t groups, float eps,
                             float* fold_ws, int in_silu, void* y, int64_t Cys, int64_t B, int64_t H,
                             int64_t W, int64_t Cout, int64_t ksize, float out_scale, double* stats,
                             brv_stream_t stream);
/* Round 6: the same two convolutions with a caller-provided scratch. Launches with few tiles (the inner levels of
 * the U-Net: up to 128 (4-row tile, 128-channel block) pairs) split the REDUCTION over workgroups -- partial sums in
 * fp32 through `split_ws`, added in split order by a second launch together with bias, residual, scale and the
 * GroupNorm statistics (csrc/conv_nhwc_splitk.cuh) -- instead of leaving most of the chip idle.
 * brv_conv_nhwc_split_ws_bytes: bytes such a launch uses (0: the launch does not split; any smaller scratch, or
 * NULL, selects the pixel-parallel kernel of the entry points above). */
int64_t brv_conv_nhwc_split_ws_bytes(int64_t B, int64_t H, int64_t W, int64_t C1, int64_t C2, int64_t Cout);
int brv_conv_nhwc_forward_ws(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                             int64_t C2s, const void* wp, const float* bias, const void* res,
                             int64_t Crs, const float* in_scale, const float* in_shift, int in_silu,
                             void* y, int64_t Cys, int64_t B, int64_t H, int64_t W, int64_t Cout,
                             int64_t ksize, float out_scale, double* stats, void* split_ws,
                             int64_t split_ws_bytes, brv_stream_t stream);
int brv_conv_nhwc_forward_gn_ws(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                                int64_t C2s, const void* wp, const float* bias, const void* res,
                                int64_t Crs, const double* sums1, const double* sums2,
                                const float* add_bc, const float* gamma, const float* beta,
                                const float* adm_scale, const float* adm_shift, int64_t groups, float eps,
                                float* fold_ws, int in_silu, void* y, int64_t Cys, int64_t B, int64_t H,
                                int64_t W, int64_t Cout, int64_t ksize, float out_scale, double* stats,
                                void* split_ws, int64_t split_ws_bytes, brv_stream_t stream);
int brv_groupnorm_fold_chan2(const double* sums1, int64_t C1, const double* sums2, int64_t C2,
                             const float* add_bc, const float* gamma, const float* beta,
                             const float* adm_scale, const float* adm_shift, float* scale_bc,
                             float* shift_bc, int64_t B, int64_t HW, int64_t groups, float eps,
                             brv_stream_t stream);
int brv_nchw_to_nhwc_f16(const float* x, void* y, int64_t B, int64_t C, int64_t Cs, int64_t HW,
                         brv_stream_t stream);
int brv_nhwc_f16_to_nchw(const void* x, float* y, int64_t B, int64_t C, int64_t Cs, int64_t HW,
                         brv_stream_t stream);
int brv_nhwc_chan_stats(const void* x, double* sums, int64_t B, int64_t C, int64_t Cs, int64_t HW,
                        int64_t c_off, int64_t Ctot, brv_stream_t stream);
int brv_groupnorm_fold_chan(const double* sums, const float* add_bc, const float* gamma,
                            const float* beta, const float* adm_scale, const float* adm_shift,
                            float* scale_bc, float* shift_bc, int64_t B, int64_t C, int64_t HW,
                            int64_t groups, float eps, brv_stream_t stream);
int brv_nhwc_affine_act(const void* x, const float* scale_bc, const float* shift_bc, void* y,
                        int64_t B, int64_t C, int64_t Cs, int64_t HW, int act, brv_stream_t stream);
int brv_nhwc_fir_resample2d(const void* x, const float* kernel, void* y, int64_t B, int64_t Cs,
                            int64_t H, int64_t W, int64_t Ho, int64_t Wo, int64_t K, int64_t pad_h,
                            int64_t pad_w, int up, float gain, brv_stream_t stream);
/* Both resamplings of a UNetBlock input in one pass (/root/reference/brever/models/sgmsep/net.py
 * UNetBlock.forward: `x = resample(x)` and `h = resample(silu(norm_1(x)))`): y_raw = FIR(x),
 * y_act = FIR(act(scale_bc*x + shift_bc)); K <= 4. */
int brv_nhwc_fir_resample2d_dual(const void* x, const float* scale_bc, const float* shift_bc, int act,
                                 const float* kernel, void* y_raw, void* y_act, int64_t B, int64_t C,
                                 int64_t Cs, int64_t H, int64_t W, int64_t Ho, int64_t Wo, int64_t K,
                                 int64_t pad_h, int64_t pad_w, int up, float gain, brv_stream_t stream);
int brv_nhwc_axpby(const void* a, float alpha, const void* b, float beta, void* out, int64_t n,
                   brv_stream_t stream);
int64_t brv_nhwc_conv1x1_packed_size(int64_t Cout, int64_t C1, int64_t C2);
int brv_nhwc_conv1x1_pack(const float* w, void* wp, int64_t Cout, int64_t C1, int64_t C2,
                          brv_stream_t stream);
int brv_nhwc_conv1x1_forward(const void* x1, int64_t C1, int64_t C1s, const void* x2, int64_t C2,
                             int64_t C2s, const void* wp, const float* bias, void* y, int64_t Cys,
                             int64_t npx, int64_t Cout, float out_scale, brv_stream_t stream);
int brv_nhwc_conv3x3_small_pack(const float* w, void* w16, int64_t Cout, int64_t C, brv_stream_t stream);
int brv_nhwc_conv3x3_small(const void* x, const void* w16, const float* bias, const float* scale_bc,
                           const float* shift_bc, int silu, const float* y_in, float* y, int64_t B,
                           int64_t C, int64_t Cs, int64_t H, int64_t W, int64_t Cout,
                           brv_stream_t stream);
int brv_nhwc_add_pointwise(const void* x, const float* aux, const float* w, const float* bias,
                           void* y, int64_t B, int64_t C, int64_t Cs, int64_t K, int64_t HW,
                           float out_scale, brv_stream_t stream);

/* ---- pieces of MultiResYuLoss (criterion.py:135-226) ----------------------------
 * brv_apply_mask: out = x with samples >= lengths[b] zeroed (apply_mask, :229-234),
 *   x/out (B, S, L). brv_l1_*: sums[r] = sum |x - y| over n samples of row r (fp64) and
 *   dx (+)= grow[r]*sign(x - y). brv_mag_l1_*: the same on the magnitudes of complex64
 *   rows of n values: sums[r] = sum ||X| - |Y||, dX = grow[r]*sign(|X|-|Y|)*X/|X|. */
int brv_apply_mask(const float* x, const int64_t* lengths, float* out, int64_t B, int64_t S,
                   int64_t L, brv_stream_t stream);
int brv_l1_forward(const float* x, const float* y, double* sums, int64_t rows, int64_t n,
                   brv_stream_t stream);
int brv_l1_backward(const float* x, const float* y, const float* grow, float* dx, int64_t rows,
                    int64_t n, int accumulate, brv_stream_t stream);
int brv_mag_l1_forward(const float* xspec, const float* yspec, double* sums, int64_t rows,
                       int64_t n, brv_stream_t stream);
int brv_mag_l1_backward(const float* xspec, const float* yspec, const float* grow,
                        float* dxspec, int64_t rows, int64_t n, brv_stream_t stream);

/* Scale-invariant variant of MultiResYuLoss (criterion.py:207-212): x is first multiplied per
 * (item, source) row by alpha = <x, y>/(<x, x> + eps) over the samples below the item length.
 * forward: out = alpha*x (zero beyond the length), stats (rows, 2) doubles = (alpha, <x,x>+eps);
 * backward: dx from the gradient g with respect to alpha*x. */
int brv_si_scale_forward(const float* x, const float* y, const int64_t* lengths, float* out,
                         double* stats, int64_t B, int64_t S, int64_t L, float eps,
                         brv_stream_t stream);
int brv_si_scale_backward(const float* g, const float* x, const float* y, const int64_t* lengths,
                          const double* stats, float* dx, int64_t B, int64_t S, int64_t L,
                          brv_stream_t stream);

/* ---- STOI / ESTOI (brever/metrics.py:19-45,98-109; pystoi's algorithm, oracle/stoi.py) -------
 * resample_poly: y[r][m] = sum_i hpad[(m + n_pre_remove)*down - i*up] x[r][i] for
 *   m < ceil(n_in*up/down) (n_in = lengths[r], or in_stride when lengths is NULL), zero beyond:
 *   scipy.signal.resample_poly with the padded, gain-scaled filter built by the caller.
 * stoi_compact: removes the 256-sample Hann frames (hop 128) of the CLEAN signal that lie more
 *   than dyn_range dB below its loudest frame from both signals and overlap-adds the rest;
 *   geom[r] = {samples, frames, 30-frame segments, kept frames} of the compacted item;
 *   energy_scratch / kept_scratch: rows*nf_max floats / ints, nf_max >= brv_stoi_frames(max length).
 * stoi_bands: spec (rows, nf_max, ncols) = per-frame DFT rows with (re, im) interleaved for
 *   bins [bin0, bin0 + ncols/2) -> tob (rows, 15, nf_max) one-third octave magnitudes;
 *   edges (15, 2) int32 = first / one-past-last bin of every band.
 * stoi_correlate: out[r] = STOI (extended = 0, clip = 10^(15/20)) or ESTOI (extended != 0) of
 *   item r from the band magnitudes of the clean and the processed signal; items with fewer
 *   than 30 frames give 1e-5 like pystoi. partial_scratch: rows*(nf_max - 29) floats. */
int brv_resample_poly(const float* x, const float* hpad, float* y, const int64_t* lengths,
                      int64_t rows, int64_t in_stride, int64_t out_stride, int64_t up,
                      int64_t down, int64_t hpad_len, int64_t n_pre_remove, brv_stream_t stream);
int64_t brv_stoi_frames(int64_t length);
int brv_stoi_compact(const float* clean, const float* proc, const int64_t* lengths, int64_t rows,
                     int64_t stride, float* clean_out, float* proc_out, int64_t out_stride,
                     int32_t* geom, float* energy_scratch, int32_t* kept_scratch, int64_t nf_max,
                     float dyn_range, brv_stream_t stream);
int brv_stoi_bands(const float* spec, const int32_t* edges, float* tob, int64_t rows,
                   int64_t nf_max, int64_t ncols, int64_t bin0, brv_stream_t stream);
int brv_stoi_correlate(const float* tob_clean, const float* tob_proc, const int32_t* geom,
                       float* partial_scratch, float* out, int64_t rows, int64_t nf_max,
                       int extended, float clip, brv_stream_t stream);

/* ---- FLAC decoding (HOST pointers: runs in the data loader, brever/data.py:143,259-268 read the
 * dataset's audio/NNNNN_<source>.flac members through soundfile / torchaudio.info) ------------
 * flac_info: frames per channel, sample rate, channels, bits per sample from STREAMINFO.
 * flac_decode: interleaved float32 samples (frames, channels) scaled by 2^-(bps-1) into `out`
 *   (capacity in frames); returns the number of frames decoded or < 0 (malformed stream, CRC
 *   mismatch, unsupported feature). RFC 9639 subset: see csrc/flac.hip. */
int brv_flac_info(const uint8_t* data, int64_t size, int64_t* frames, int32_t* sample_rate,
                  int32_t* channels, int32_t* bits_per_sample);
int64_t brv_flac_decode(const uint8_t* data, int64_t size, float* out, int64_t capacity_frames);
/* flac_encode16 (HOST pointers): mono 16-bit FLAC stream of `frames` samples -- what
 * scripts/test_model.py --output_dir writes per signal (reference: torchaudio.save(<name>.flac),
 * scripts/test_model.py:201-209). Fixed predictors of order 0-4, partitioned Rice residuals, frame
 * CRCs. out == NULL: returns the size of the stream; else the bytes written, or < 0 (-2: capacity). */
int64_t brv_flac_encode16(const int16_t* pcm, int64_t frames, int32_t sample_rate, uint8_t* out,
                          int64_t capacity);

/* ---- optimizer --------------------------------------------------------------
 * clip_grad_norm_(max_norm) + Adam.step (base.py:296-301, torch.optim.Adam with
 * amsgrad=False, weight_decay=0) on flat buffers of n floats. grads are first
 * multiplied by grad_scale (1/world_size after a summing all-reduce). The
 * clipped gradient is written back to g. scratch: >= 16 bytes, norm_out may be
 * NULL. step is the 1-based step count. */
int brv_clip_adam_step(float* params, float* grads, float* exp_avg,
                       float* exp_avg_sq, int64_t n, float grad_scale,
                       float max_norm, float lr, float beta1, float beta2,
                       float eps, int64_t step, void* scratch, float* norm_out,
                       brv_stream_t stream);
/* The same step in two launches (no memset, no separate add): `scratch` holds two fp64 accumulators,
 * both zero before the first call; call `slot` (0 / 1, alternating between consecutive calls)
 * accumulates the squared norm into its own and the Adam kernel zeroes the other for the next call.
 * `grads2` (nullable): the weight gradient of a second kernel chain (brever_amd/models/convtasnet.py:
 * two half-batch chains), added into `grads` in the norm pass and left ZEROED for the next step. */
int brv_clip_adam_step2(float* params, float* grads, float* grads2, float* exp_avg,
                        float* exp_avg_sq, int64_t n, float grad_scale, float max_norm,
                        float lr, float beta1, float beta2, float eps, int64_t step,
                        void* scratch, int32_t slot, float* norm_out, brv_stream_t stream);
/* hipMemsetAsync(ptr, 0, bytes) and the mean of n <= 2^20 floats (per-item losses -> the step's
 * loss): the host's training step issues no PyTorch kernel between forward and optimizer. */
int brv_memset_zero(void* ptr, int64_t bytes, brv_stream_t stream);
int brv_mean_f32(const float* x, int64_t n, float* out, brv_stream_t stream);

/* Exponential moving average of the parameters (EMA / EMAKarras.update,
 * brever/modules/ema.py:36-39): ema += (1 - beta)*(param - ema), rounded as the reference. */
int brv_ema_update(float* ema, const float* param, float one_minus_beta, int64_t n,
                   brv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BREVER_HIP_H */
