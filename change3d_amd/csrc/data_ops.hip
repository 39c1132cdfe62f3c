// On-device input pipeline of the BCD train step: what the reference does per sample on the
// host with numpy / cv2 before the tensors ever reach the GPU (reference data/transforms.py:100-154:
// random_flip -> random_exchange -> normalize -> to_tensor, composed at scripts/train_BCD.py:262-270) runs here
// as ONE pass over the raw uint8 batch already resident in HBM:
//
//   image u8 [B][H][W][6] (pre RGB | post RGB, the reference's 6-channel HWC array), label u8 [B][H][W]
//   flags u8 [B][3] = (flip around the x axis = cv2.flip(.,0), flip around the y axis = cv2.flip(.,1), exchange)
//   -> pre f32 [B][3][H][W], post f32 [B][3][H][W] = ((u8 / 255) - mean) / std   (f32, IEEE division: bit-identical
//      to numpy's `image.astype(float32) / 255.0`, `(image - mean) / std`),  label f32 [B][1][H][W] = ceil(u8/255).
//
// HBM-bound byte shuffling: 7 B read, 28 B written per pixel; each thread owns 4 consecutive output pixels of one
// row so that the f32 stores are 16-byte vectors (the u8 reads of a wave cover a contiguous 6 KB span).
#include "common.h"
#include "../../include/change3d_hip.h"

namespace {

__global__ __launch_bounds__(256) void bcd_preprocess_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lab,
                                                             const uint8_t* __restrict__ flags, const float* __restrict__ mean,
                                                             const float* __restrict__ stdv, float* __restrict__ pre,
                                                             float* __restrict__ post, float* __restrict__ label, int B, int H,
                                                             int W) {
  const int wq = (W + 3) >> 2;                                    // 4-pixel groups per row
  const int64_t total = (int64_t)B * H * wq;
  float m[6], s[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) { m[c] = mean[c]; s[c] = stdv[c]; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % wq);
    const int64_t r = i / wq;
    const int y = (int)(r % H), b = (int)(r / H);
    const bool fy = flags && flags[b * 3 + 0], fx = flags && flags[b * 3 + 1], ex = flags && flags[b * 3 + 2];
    const int ys = fy ? H - 1 - y : y;                            // source row
    float o[6][4];
    float l[4];
    const int x0 = q * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int x = x0 + k;
      const int xs = x < W ? (fx ? W - 1 - x : x) : 0;
      const uint8_t* p = img + (((int64_t)b * H + ys) * W + xs) * 6;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        // channel c of the OUTPUT 6-channel image: after an exchange the first three come from post
        const int cs = ex ? (c < 3 ? c + 3 : c - 3) : c;
        const float v = (float)p[cs] / 255.0f;
        o[c][k] = (v - m[c]) / s[c];
      }
      l[k] = lab ? (lab[((int64_t)b * H + ys) * W + xs] ? 1.0f : 0.0f) : 0.0f;
    }
    const int64_t plane = (int64_t)H * W;
    const int64_t base = (int64_t)y * W + x0;
    if (x0 + 3 < W && (W & 3) == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        *reinterpret_cast<float4*>(pre + ((int64_t)b * 3 + c) * plane + base) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
        *reinterpret_cast<float4*>(post + ((int64_t)b * 3 + c) * plane + base) = make_float4(o[c + 3][0], o[c + 3][1], o[c + 3][2], o[c + 3][3]);
      }
      if (label) *reinterpret_cast<float4*>(label + (int64_t)b * plane + base) = make_float4(l[0], l[1], l[2], l[3]);
    } else {
      for (int k = 0; k < 4 && x0 + k < W; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          pre[((int64_t)b * 3 + c) * plane + base + k] = o[c][k];
          post[((int64_t)b * 3 + c) * plane + base + k] = o[c + 3][k];
        }
        if (label) label[(int64_t)b * plane + base + k] = l[k];
      }
    }
  }
}

// clip[b][c][t][y][x] (f32 NCDHW, what the stem reads): t = 0 <- pre, t = 1..K <- perception frames (shared by the
// batch), t = K+1 <- post.  One 16-byte vector per thread per step; replaces expand + torch.cat
// (reference model/trainer.py:155-162).
__global__ __launch_bounds__(256) void build_clip_kernel(const float* __restrict__ pre, const float* __restrict__ post,
                                                         const float* __restrict__ P, float* __restrict__ clip, int B, int K,
                                                         int64_t hw4) {
  const int T = K + 2;
  const int64_t total = (int64_t)B * 3 * T * hw4;
  const float4* p4 = reinterpret_cast<const float4*>(pre);
  const float4* q4 = reinterpret_cast<const float4*>(post);
  const float4* f4 = reinterpret_cast<const float4*>(P);
  float4* o4 = reinterpret_cast<float4*>(clip);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i % hw4;
    int64_t q = i / hw4;
    const int t = (int)(q % T); q /= T;
    const int c = (int)(q % 3);
    const int b = (int)(q / 3);
    float4 v;
    if (t == 0) v = p4[((int64_t)b * 3 + c) * hw4 + r];
    else if (t == T - 1) v = q4[((int64_t)b * 3 + c) * hw4 + r];
    else v = f4[((int64_t)c * K + (t - 1)) * hw4 + r];
    o4[i] = v;
  }
}

}  // namespace

extern "C" int c3d_build_clip(const float* pre, const float* post, const float* frames, float* clip, int32_t B, int32_t K,
                              int32_t H, int32_t W, void* stream) {
  if (!pre || !post || !frames || !clip || B <= 0 || K <= 0 || H <= 0 || W <= 0) return C3D_E_BADARG;
  if (((int64_t)H * W) & 3) return C3D_E_UNSUPPORTED;
  const int64_t hw4 = (int64_t)H * W / 4, total = (int64_t)B * 3 * (K + 2) * hw4;
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  build_clip_kernel<<<dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(pre, post, frames, clip, B,
                                                                                                  K, hw4);
  C3D_CHECK_LAUNCH();
  return 0;
}

// SCD labels (reference data/transforms.py:300-326 SCDTransforms.random_flip / random_exchange, :341-357 to_tensor;
// scripts/train_SCD.py:207-213 `.long()`): label u8 [B][H][W][3] = (pre classes, post classes, change) with the same
// flip / exchange flags as the image (c3d_bcd_preprocess handles the image: the 6-channel arithmetic is identical)
// -> int64 [B][3][H][W]; an exchange swaps the two class maps.
__global__ __launch_bounds__(256) void scd_label_kernel(const uint8_t* __restrict__ lab, const uint8_t* __restrict__ flags,
                                                        int64_t* __restrict__ out, int B, int H, int W) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int64_t r = i / W;
    const int y = (int)(r % H), b = (int)(r / H);
    const bool fy = flags && flags[b * 3 + 0], fx = flags && flags[b * 3 + 1], ex = flags && flags[b * 3 + 2];
    const uint8_t* p = lab + (((int64_t)b * H + (fy ? H - 1 - y : y)) * W + (fx ? W - 1 - x : x)) * 3;
    const int64_t plane = (int64_t)H * W, o = (int64_t)b * 3 * plane + (int64_t)y * W + x;
    out[o] = p[ex ? 1 : 0];
    out[o + plane] = p[ex ? 0 : 1];
    out[o + 2 * plane] = p[2];
  }
}

// Change-captioning image pairs (reference data/dataset.py:411-424 CaptionDataset.__getitem__ + scripts/train_CC.py:466-469
// transforms.Normalize): img u8 [B][2][3][H][W] (planar, as the HDF5 file stores them) -> pre, post f32 [B][3][H][W] =
// Normalize(FloatTensor(u8 / 255.)) through a 3 x 256 table built by the HOST with exactly that arithmetic (f64 division,
// rounded to f32, then f32 sub / div), so every value is bit-identical to the reference's; swap[b] != 0 exchanges the
// pair (the TRAIN split's p = 0.3 augmentation).  8 B read / 32 B written per 4 pixels and plane.
__global__ __launch_bounds__(256) void cc_preprocess_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ swap,
                                                            const float* __restrict__ lut, float* __restrict__ pre,
                                                            float* __restrict__ post, int B, int64_t hw4) {
  __shared__ float tab[3 * 256];
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) tab[i] = lut[i];
  __syncthreads();
  const int64_t total = (int64_t)B * 6 * hw4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = i % hw4;
    const int pc = (int)((i / hw4) % 6), b = (int)(i / (hw4 * 6));   // pc = image (0 / 1) * 3 + channel
    const uchar4 v = reinterpret_cast<const uchar4*>(img)[i];
    const float* t = tab + (pc % 3) * 256;
    const bool second = (pc >= 3) != (swap && swap[b]);
    float* dst = (second ? post : pre) + ((int64_t)b * 3 + pc % 3) * hw4 * 4 + q * 4;
    *reinterpret_cast<float4*>(dst) = make_float4(t[v.x], t[v.y], t[v.z], t[v.w]);
  }
}

extern "C" int c3d_scd_label_preprocess(const uint8_t* label3, const uint8_t* flags, int64_t* out, int32_t B, int32_t H,
                                        int32_t W, void* stream) {
  if (!label3 || !out || B <= 0 || H <= 0 || W <= 0) return C3D_E_BADARG;
  int64_t grid = ((int64_t)B * H * W + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  scd_label_kernel<<<dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(label3, flags, out, B, H, W);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_cc_preprocess(const uint8_t* img, const uint8_t* swap, const float* lut, float* pre, float* post,
                                 int32_t B, int32_t H, int32_t W, void* stream) {
  if (!img || !lut || !pre || !post || B <= 0 || H <= 0 || W <= 0 || (((int64_t)H * W) & 3)) return C3D_E_BADARG;
  const int64_t hw4 = (int64_t)H * W / 4;
  int64_t grid = ((int64_t)B * 6 * hw4 + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  cc_preprocess_kernel<<<dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(img, swap, lut, pre, post,
                                                                                                     B, hw4);
  C3D_CHECK_LAUNCH();
  return 0;
}

extern "C" int c3d_bcd_preprocess(const uint8_t* image6, const uint8_t* label, const uint8_t* flags, const float* mean6,
                                  const float* std6, float* pre, float* post, float* label_out, int32_t B, int32_t H,
                                  int32_t W, void* stream) {
  if (!image6 || !mean6 || !std6 || !pre || !post || B <= 0 || H <= 0 || W <= 0) return C3D_E_BADARG;
  if ((label == nullptr) != (label_out == nullptr)) return C3D_E_BADARG;
  const int64_t total = (int64_t)B * H * ((W + 3) / 4);
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  bcd_preprocess_kernel<<<dim3((unsigned)grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream)>>>(
      image6, label, flags, mean6, std6, pre, post, label_out, B, H, W);
  C3D_CHECK_LAUNCH();
  return 0;
}
