// input.hip -- the training input pipeline (transform_tr) on the GPU, bit-exact with the PIL calls the reference makes
// (main.py:409-419 transform_tr; dataloaders.py:139-150 RandomHorizontalFlip, 398-435 RandomSizeAndCrop -> img.resize(BICUBIC)
// / mask.resize(NEAREST), 257-337 RandomCrop with ImageOps.expand padding, 467-482 Resize, 118-136 ToTensor).
//
// PIL (Pillow 12.2, src/libImaging/Resample.c) resizes 8-bit images in two separable passes with fixed-point
// coefficients: out = clip8((2^21 + sum_x in[xmin + x] * k[x]) >> 22), horizontal pass first, 8-bit intermediate.  The
// coefficient / bounds tables are built on the host in double precision exactly as Pillow does (mrfp_amd/input_pipeline.py,
// oracle/input_oracle.py); the kernels do the integer arithmetic, so the result equals PIL's byte for byte.
//   resample_u8   one separable pass over a [H,W,C] uint8 image (the horizontal pass can read the source mirrored: the
//                 reference flips BEFORE it scales)
//   box_blur3     one pass of ImageFilter.GaussianBlur's box-blur approximation (radius < 1: dataloaders.py:168-177)
//   assemble      pad (ImageOps.expand: image 0, label ignore_index) + crop + ToTensor: float32 [3,Hc,Wc] in 0..255 and the
//                 int64 label map, the label fetched through PIL's nearest-neighbour index tables from the ORIGINAL map
#include "common.hpp"

namespace mrfp {

constexpr int kPrecisionBits = 32 - 8 - 2;      // Pillow Resample.c PRECISION_BITS

__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int Hin,
                                                          int Win, int Hout, int Wout, int C, const int32_t* __restrict__ bounds,
                                                          const int32_t* __restrict__ coefs, int ksize, int vertical, int flip) {
    const int64_t n = (int64_t)Hout * Wout * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int ox = (int)(pix % Wout), oy = (int)(pix / Wout);
        const int o = vertical ? oy : ox;
        const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int32_t* k = coefs + (int64_t)o * ksize;
        int acc = 1 << (kPrecisionBits - 1);
        if (vertical) {
            for (int t = 0; t < cnt; ++t) acc += (int)src[((int64_t)(lo + t) * Win + ox) * C + c] * k[t];
        } else {
            for (int t = 0; t < cnt; ++t) {
                const int sx = flip ? Win - 1 - (lo + t) : lo + t;
                acc += (int)src[((int64_t)oy * Win + sx) * C + c] * k[t];
            }
        }
        const int v = acc >> kPrecisionBits;                // arithmetic shift, then clip8
        dst[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

__global__ __launch_bounds__(256) void input_assemble_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lab,
                                                             const int32_t* __restrict__ ytab, const int32_t* __restrict__ xtab,
                                                             int Hs, int Ws, int Hl, int Wl, int flip, int pad_x, int pad_y, int x1,
                                                             int y1, int Hc, int Wc, int ignore, float* __restrict__ out_img,
                                                             uint8_t* __restrict__ out_u8, int64_t* __restrict__ out_lab) {
    const int64_t n = (int64_t)Hc * Wc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % Wc), oy = (int)(i / Wc);
        const int px = x1 + ox - pad_x, py = y1 + oy - pad_y;       // position in the scaled image
        const bool inside = px >= 0 && px < Ws && py >= 0 && py < Hs;
        uint8_t r = 0, g = 0, b = 0;
        int64_t l = ignore;
        if (inside) {
            const uint8_t* p = img + ((int64_t)py * Ws + px) * 3;
            r = p[0]; g = p[1]; b = p[2];
            const int sy = ytab[py], sx0 = xtab[px];
            if (sy >= 0 && sy < Hl && sx0 >= 0 && sx0 < Wl) l = lab[(int64_t)sy * Wl + (flip ? Wl - 1 - sx0 : sx0)];
            else l = 0;                                               // ImagingScaleAffine leaves such pixels of a new image zero
        }
        if (out_u8) {                                   // uint8 [Hc,Wc,3]: the blur passes come before ToTensor
            out_u8[3 * i] = r; out_u8[3 * i + 1] = g; out_u8[3 * i + 2] = b;
        } else {
            out_img[i] = (float)r;
            out_img[n + i] = (float)g;
            out_img[2 * n + i] = (float)b;
        }
        out_lab[i] = l;
    }
}

// One pass of Pillow's box blur (BoxBlur.c ImagingLineBoxBlur8) for a box radius below 1 -- all ImageFilter.GaussianBlur
// ever asks for when its radius is random.random() (dataloaders.py:172-174): out = (in*ww + (left + right)*fw + 2^23) >> 24
// with the edge pixels replicated; GaussianBlur = 3 horizontal + 3 vertical passes, every pass rounded to 8 bits.
__global__ __launch_bounds__(256) void box_blur3_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W,
                                                           int C, unsigned ww, unsigned fw, int vertical) {
    const int64_t n = (int64_t)H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int x = (int)(pix % W), y = (int)(pix / W);
        unsigned a, b;
        if (vertical) {
            a = src[((int64_t)(y > 0 ? y - 1 : 0) * W + x) * C + c];
            b = src[((int64_t)(y < H - 1 ? y + 1 : H - 1) * W + x) * C + c];
        } else {
            a = src[((int64_t)y * W + (x > 0 ? x - 1 : 0)) * C + c];
            b = src[((int64_t)y * W + (x < W - 1 ? x + 1 : W - 1)) * C + c];
        }
        const unsigned bulk = (unsigned)src[i] * ww + (a + b) * fw;
        dst[i] = (uint8_t)((bulk + (1u << 23)) >> 24);
    }
}

__global__ __launch_bounds__(256) void u8hwc_to_f32chw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t npix) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        dst[i] = (float)src[3 * i];
        dst[npix + i] = (float)src[3 * i + 1];
        dst[2 * npix + i] = (float)src[3 * i + 2];
    }
}

// ---- ColorJitter (dataloaders.py:491-660): PIL ImageEnhance blends and the HSV round trip, per pixel ---------------------
// ImageEnhance.X(img).enhance(f) = Image.blend(degenerate, img, f) = clip8((float)d + f * (float)(p - d)) per byte (Blend.c,
// C float arithmetic), with d = 0 (Brightness), the rounded mean of the L image (Contrast) or the pixel's own L (Color);
// L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16 (Convert.c).  adjust_hue: RGB -> HSV (Convert.c rgb2hsv: float / double
// mix as in the C source), H += shift (uint8 wrap), HSV -> RGB.  Restated in oracle/input_oracle.py and pinned against PIL
// there (the two conversions on all 2^24 triples).
__device__ __forceinline__ int pil_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
__device__ __forceinline__ uint8_t pil_blend(int d, int p, float alpha) {
    const float t = (float)d + alpha * (float)(p - d);
    return (uint8_t)(t <= 0.f ? 0 : t >= 255.f ? 255 : (int)t);
}

__global__ __launch_bounds__(256) void gray_sum_kernel(const uint8_t* __restrict__ src, int64_t npix, unsigned long long* __restrict__ sum) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256)
        acc += (unsigned)pil_l(src[3 * i], src[3 * i + 1], src[3 * i + 2]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);          // integer sum: exact in any order
}
__global__ void gray_mean_kernel(const unsigned long long* __restrict__ sum, int64_t npix, int* __restrict__ gray) {
    *gray = (int)((double)*sum / (double)npix + 0.5);                  // int(ImageStat.Stat(L).mean[0] + 0.5)
}

// op: 0 brightness, 1 contrast (*gray), 2 saturation (own L), 3 hue (shift)
__global__ __launch_bounds__(256) void jitter_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t npix, int op,
                                                        float alpha, int shift, const int* __restrict__ gray) {
    const int gm = (op == 1) ? *gray : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const int r = src[3 * i], g = src[3 * i + 1], b = src[3 * i + 2];
        uint8_t o0, o1, o2;
        if (op < 3) {
            const int d = op == 0 ? 0 : op == 1 ? gm : pil_l(r, g, b);
            o0 = pil_blend(d, r, alpha); o1 = pil_blend(d, g, alpha); o2 = pil_blend(d, b, alpha);
        } else {
            const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
            int uh = 0, us = 0;
            const int uv = maxc;
            if (minc != maxc) {
                const float cr = (float)(maxc - minc);
                const float sf = cr / (float)maxc;
                const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
                float h;
                if (r == maxc) h = bc - gc;
                else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
                else h = (float)(4.0 + (double)gc - (double)rc);
                const double hd = (double)h / 6.0 + 1.0;
                h = (float)(hd - floor(hd));                           // fmod(., 1.0) of a positive number
                uh = (int)((double)h * 255.0);
                us = (int)((double)sf * 255.0);
                uh = uh < 0 ? 0 : uh > 255 ? 255 : uh;
                us = us < 0 ? 0 : us > 255 ? 255 : us;
            }
            uh = (uh + shift) & 255;                                   // np_h += np.uint8(hue_factor * 255)
            if (us == 0) {
                o0 = o1 = o2 = (uint8_t)uv;
            } else {
                const float fs = (float)us / 255.0f;
                const double hh = (double)uh * 6.0 / 255.0;
                const int ii = (int)floor(hh);
                const float f = (float)(hh - (double)ii);
                const float vf = (float)uv;
                const double pd = round((double)(vf * (1.0f - fs))), qd = round((double)(vf * (1.0f - fs * f))),
                             td = round((double)(vf * (1.0f - fs * (1.0f - f))));
                const uint8_t pp = (uint8_t)(pd < 0 ? 0 : pd > 255 ? 255 : pd), qq = (uint8_t)(qd < 0 ? 0 : qd > 255 ? 255 : qd),
                              tt = (uint8_t)(td < 0 ? 0 : td > 255 ? 255 : td), vv = (uint8_t)uv;
                switch (ii % 6) {
                    case 0: o0 = vv; o1 = tt; o2 = pp; break;
                    case 1: o0 = qq; o1 = vv; o2 = pp; break;
                    case 2: o0 = pp; o1 = vv; o2 = tt; break;
                    case 3: o0 = pp; o1 = qq; o2 = vv; break;
                    case 4: o0 = tt; o1 = pp; o2 = vv; break;
                    default: o0 = vv; o1 = pp; o2 = qq; break;
                }
            }
        }
        dst[3 * i] = o0; dst[3 * i + 1] = o1; dst[3 * i + 2] = o2;
    }
}

}  // namespace mrfp

using namespace mrfp;

extern "C" {

int mrfp_resample_u8(const void* src, void* dst, int64_t Hin, int64_t Win, int64_t Hout, int64_t Wout, int64_t C,
                     const int32_t* bounds, const int32_t* coefs, int ksize, int vertical, int flip, void* stream) {
    MRFP_CHECK(src && dst && bounds && coefs && ksize > 0, "resample_u8: null argument");
    MRFP_CHECK(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0 && Hin < 65536 && Win < 65536 && Hout < 65536 && Wout < 65536,
               "resample_u8: bad sizes");
    MRFP_CHECK(vertical ? Wout == Win : Hout == Hin, "resample_u8: a pass changes one axis only (%s pass: %lldx%lld -> %lldx%lld)",
               vertical ? "vertical" : "horizontal", (long long)Hin, (long long)Win, (long long)Hout, (long long)Wout);
    MRFP_CHECK(!(vertical && flip), "resample_u8: the mirrored read belongs to the horizontal pass");
    const int64_t n = Hout * Wout * C;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(resample_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src,
                       (uint8_t*)dst, (int)Hin, (int)Win, (int)Hout, (int)Wout, (int)C, bounds, coefs, ksize, vertical, flip);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_input_assemble(const void* img, const void* lab, const int32_t* ytab, const int32_t* xtab, int64_t Hs, int64_t Ws,
                        int64_t Hl, int64_t Wl, int flip, int pad_x, int pad_y, int x1, int y1, int64_t Hc, int64_t Wc, int ignore,
                        float* out_img, void* out_u8, int64_t* out_lab, void* stream) {
    MRFP_CHECK(img && lab && ytab && xtab && (out_img || out_u8) && out_lab, "input_assemble: null argument");
    MRFP_CHECK(Hs > 0 && Ws > 0 && Hl > 0 && Wl > 0 && Hc > 0 && Wc > 0 && Hs < 65536 && Ws < 65536 && Hc < 65536 && Wc < 65536,
               "input_assemble: bad sizes");
    MRFP_CHECK(pad_x >= 0 && pad_y >= 0 && x1 >= 0 && y1 >= 0 && x1 + Wc <= Ws + 2 * pad_x && y1 + Hc <= Hs + 2 * pad_y,
               "input_assemble: the crop [%d,%d)+%lldx%lld leaves the padded image %lldx%lld", x1, y1, (long long)Wc, (long long)Hc,
               (long long)(Ws + 2 * pad_x), (long long)(Hs + 2 * pad_y));
    const int64_t n = Hc * Wc;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(input_assemble_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img,
                       (const uint8_t*)lab, ytab, xtab, (int)Hs, (int)Ws, (int)Hl, (int)Wl, flip, pad_x, pad_y, x1, y1, (int)Hc, (int)Wc,
                       ignore, out_img, (uint8_t*)out_u8, out_lab);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_box_blur3_u8(const void* src, void* dst, int64_t H, int64_t W, int64_t C, int64_t ww, int64_t fw, int vertical,
                      void* stream) {
    MRFP_CHECK(src && dst && src != dst && H > 0 && W > 0 && C > 0 && H < 65536 && W < 65536, "box_blur3_u8: bad arguments");
    MRFP_CHECK(ww > 0 && fw >= 0 && ww + 2 * fw <= (1ll << 24), "box_blur3_u8: weights %lld + 2*%lld exceed 2^24 (box radius >= 1?)",
               (long long)ww, (long long)fw);
    const int64_t n = H * W * C;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(box_blur3_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src,
                       (uint8_t*)dst, (int)H, (int)W, (int)C, (unsigned)ww, (unsigned)fw, vertical);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_jitter_u8(const void* src, void* dst, int64_t npix, int op, float factor, int shift, void* ws, void* stream) {
    MRFP_CHECK(src && dst && npix > 0 && op >= 0 && op <= 3, "jitter_u8: bad arguments (op 0..3)");
    MRFP_CHECK(op != 1 || ws, "jitter_u8: the contrast op needs 16 bytes of workspace");
    MRFP_CHECK(op != 3 || (shift >= 0 && shift < 256), "jitter_u8: hue shift must be in 0..255");
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = (npix + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    int* gray = nullptr;
    if (op == 1) {
        unsigned long long* sum = (unsigned long long*)ws;
        gray = (int*)((char*)ws + 8);
        if (hipMemsetAsync(sum, 0, 8, st) != hipSuccess) { set_error("jitter_u8: hipMemsetAsync failed"); return -1; }
        hipLaunchKernelGGL(gray_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint8_t*)src, npix, sum);
        hipLaunchKernelGGL(gray_mean_kernel, dim3(1), dim3(1), 0, st, (const unsigned long long*)sum, npix, gray);
    }
    hipLaunchKernelGGL(jitter_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint8_t*)src, (uint8_t*)dst, npix, op, factor,
                       shift, (const int*)gray);
    MRFP_LAUNCH_CHECK();
    return 0;
}

int mrfp_u8hwc_to_f32chw(const void* src, float* dst, int64_t H, int64_t W, void* stream) {
    MRFP_CHECK(src && dst && H > 0 && W > 0, "u8hwc_to_f32chw: bad arguments");
    const int64_t n = H * W;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(u8hwc_to_f32chw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, dst, n);
    MRFP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
