// Host image path on the device, the part that can be pinned, gfx950.
//
// Reference: code/lib/utils/blob.py:34-79 (prep_im_for_blob), :19-32 (im_list_to_blob),
// roi_data_layer/minibatch_bus.py:269-272 (one grey plane stacked three times, horizontal flip),
// datasets/imdb.py:106-121 (box mirroring of flipped images).  skimage.transform.resize
// (blob.py:74-77) sits in the middle of prep_im_for_blob; the fixtures made with the reference's own
// blob.py pin the steps around it, the resize itself follows the published algorithm of the pinned
// release (scikit-image 0.14.2, README.md:13; library absent: "parity unpinned", oracle/np_oracle.py):
//   wssdl_image_prep     u8 grey plane -> f32 [h,w,3]: flip, /255, brightness (+delta, clip),
//                        contrast ((x - mean) * f + mean, clip), - pixel_mean/255   (what the
//                        reference hands to the resize); one pass, 1 B read + 12 B written per pixel
//   wssdl_image_resize   skimage.transform.resize(im, [rows, cols]) with the 0.14 defaults: order 1,
//   wssdl_image_warp     mode 'constant', cval 0, clip to the input's range, no anti-aliasing = warp() with
//                        the pixel-centre affine map; the general form takes any 3x3 inverse map (rotate())
//   wssdl_image_to_blob  the resize's f64 (or f32) output -> / (pixel_std/255) or * 255, placed
//                        zero-padded into blob[i] (what im_list_to_blob builds)
//   wssdl_flip_boxes     x1' = width - x2 - 1, x2' = width - x1 - 1
// Arithmetic follows NumPy's: f32 for the augmentation (the random draws are Python floats:
// weak scalars), f64 for the mean / std steps, each rounded to f32 once.  The contrast step needs
// the image mean: NumPy sums f32 pairwise; here the sum is exact-ish (f64, fixed order), so the
// mean -- and with it the output -- can differ from NumPy's in the last bit (tests: 2e-7 abs).
#include "common.hip.h"

namespace wssdl {

constexpr int IMG_PARTS = 256;      // partial sums of the mean

__device__ __forceinline__ float prep_value(unsigned char u, int use_b, float delta) {
    float v = (float)u / 255.0f;
    if (use_b) {
        v = v + delta;
        v = fminf(fmaxf(v, 0.0f), 1.0f);
    }
    return v;
}

// partial f64 sums of the brightened plane: block b sums pixels b, b + IMG_PARTS*256, ... (fixed order)
__global__ __launch_bounds__(256) void image_sum_kernel(const unsigned char *__restrict__ gray, int h, int w,
                                                        int row_stride, int use_b, float delta,
                                                        double *__restrict__ parts) {
    __shared__ double s[256];
    const long long n = (long long)h * w;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)IMG_PARTS * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        acc += (double)prep_value(gray[(size_t)y * row_stride + x], use_b, delta);
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) parts[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void image_prep_kernel(const unsigned char *__restrict__ gray, int h, int w,
                                                         int row_stride, int flipped, int use_b, float delta,
                                                         int use_c, float factor, double mean_sub,
                                                         const double *__restrict__ parts, float *__restrict__ out) {
    __shared__ double s[256];
    float mm = 0.0f;
    if (use_c) {                       // every block re-reduces the partial sums in the same order
        s[threadIdx.x] = threadIdx.x < IMG_PARTS ? parts[threadIdx.x] : 0.0;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
            __syncthreads();
        }
        mm = (float)(s[0] / (double)((long long)h * w));
    }
    const long long n = (long long)h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        const int sx = flipped ? w - 1 - x : x;                 // im[:, ::-1, :]
        float v = prep_value(gray[(size_t)y * row_stride + sx], use_b, delta);
        if (use_c) {
            v = v - mm;
            v = v * factor;
            v = v + mm;
            v = fminf(fmaxf(v, 0.0f), 1.0f);
        }
        const float o = (float)((double)v - mean_sub);          // im -= pixel_means / 255.
        float *p = out + (size_t)i * 3;
        p[0] = o;  p[1] = o;  p[2] = o;
    }
}

// ---- blob.py:49-60 on the float64 image skimage.transform.rotate returns (weak images, :39-41): the
// same steps as above, all in f64 (NumPy: a float64 array and Python-float scalars).  The input is a
// strided view [h, w, C] (the crop :43-47 is a slice), elements (y, x, ch) at im[y*row_stride + x*C + ch].
__device__ __forceinline__ double adjust_value(double v, int use_b, double delta) {
    if (use_b) {
        v = v + delta;
        v = fmin(fmax(v, 0.0), 1.0);
    }
    return v;
}

__global__ __launch_bounds__(256) void image_sum_f64_kernel(const double *__restrict__ im, int h, int wc,
                                                            long long row_stride, int use_b, double delta,
                                                            double *__restrict__ parts) {
    __shared__ double s[256];
    const long long n = (long long)h * wc;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)IMG_PARTS * 256) {
        const int y = (int)(i / wc), x = (int)(i - (long long)y * wc);
        acc += adjust_value(im[(size_t)y * row_stride + x], use_b, delta);
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) parts[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void image_adjust_f64_kernel(const double *__restrict__ im, int h, int wc,
                                                               long long row_stride, int use_b, double delta,
                                                               int use_c, double factor, double mean_sub,
                                                               const double *__restrict__ parts,
                                                               double *__restrict__ out) {
    __shared__ double s[256];
    double mm = 0.0;
    if (use_c) {
        s[threadIdx.x] = threadIdx.x < IMG_PARTS ? parts[threadIdx.x] : 0.0;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
            __syncthreads();
        }
        mm = s[0] / (double)((long long)h * wc);
    }
    const long long n = (long long)h * wc;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / wc), x = (int)(i - (long long)y * wc);
        double v = adjust_value(im[(size_t)y * row_stride + x], use_b, delta);
        if (use_c) {
            v = v - mm;
            v = v * factor;
            v = v + mm;
            v = fmin(fmax(v, 0.0), 1.0);
        }
        out[i] = v - mean_sub;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void image_to_blob_kernel(const T *__restrict__ im, int h, int w, double scale,
                                                            int divide, float *__restrict__ blob_i, int Hmax,
                                                            int Wmax) {
    const long long n = (long long)Hmax * Wmax * 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % 3);
        const long long px = i / 3;
        const int y = (int)(px / Wmax), x = (int)(px - (long long)y * Wmax);
        float o = 0.0f;
        if (y < h && x < w) {
            const double v = (double)im[((size_t)y * w + x) * 3 + c];
            o = (float)(divide ? v / scale : v * scale);
        }
        blob_i[i] = o;
    }
}

// ---- skimage.transform.warp, order 1 (skimage/transform/_warps_cy.pyx `_warp_fast`, interpolation.pxd
// `bilinear_interpolation` / `get_pixel2d`, _warps.py `_clip_warp_output`) -------------------------------
constexpr int WARP_PARTS = 256;
struct WarpMatrix { double m[9]; };

// partial min / max of the input in f64 (np.min / np.max of the whole array, all channels): block b covers
// elements b*256 + t, stride WARP_PARTS*256
template <typename T>
__global__ __launch_bounds__(256) void image_minmax_kernel(const T *__restrict__ im, long long n,
                                                           double *__restrict__ parts) {
    __shared__ double lo_s[256], hi_s[256];
    double lo = INFINITY, hi = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)WARP_PARTS * 256) {
        const double v = (double)im[i];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    lo_s[threadIdx.x] = lo;  hi_s[threadIdx.x] = hi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            lo_s[threadIdx.x] = fmin(lo_s[threadIdx.x], lo_s[threadIdx.x + st]);
            hi_s[threadIdx.x] = fmax(hi_s[threadIdx.x], hi_s[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { parts[blockIdx.x] = lo_s[0];  parts[WARP_PARTS + blockIdx.x] = hi_s[0]; }
}

template <typename T>
__device__ __forceinline__ double warp_pixel(const T *__restrict__ im, int h, int w, int C, int ch, long long r,
                                             long long c, int mode, double cval) {
    if (mode == WSSDL_WARP_EDGE) {
        r = r < 0 ? 0 : (r >= h ? h - 1 : r);
        c = c < 0 ? 0 : (c >= w ? w - 1 : c);
    } else if (r < 0 || r >= h || c < 0 || c >= w) {
        return cval;                                            // get_pixel2d, mode 'C'
    }
    return (double)im[((size_t)r * w + (size_t)c) * C + ch];
}

// one thread per output pixel, all channels (the grey plane x3 shares the four source cells)
template <typename T>
__global__ __launch_bounds__(256) void image_warp_kernel(const T *__restrict__ im, int h, int w, int C, WarpMatrix M,
                                                         int rows, int cols, int mode, double cval, int clip,
                                                         const double *__restrict__ parts, double *__restrict__ out) {
    __shared__ double lo_s[256], hi_s[256];
    double lo = 0.0, hi = 0.0;
    bool keep_cval = false;
    if (clip) {                                                 // every block re-reduces the partials
        lo_s[threadIdx.x] = parts[threadIdx.x];
        hi_s[threadIdx.x] = parts[WARP_PARTS + threadIdx.x];
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) {
                lo_s[threadIdx.x] = fmin(lo_s[threadIdx.x], lo_s[threadIdx.x + st]);
                hi_s[threadIdx.x] = fmax(hi_s[threadIdx.x], hi_s[threadIdx.x + st]);
            }
            __syncthreads();
        }
        lo = lo_s[0];  hi = hi_s[0];
        keep_cval = mode == WSSDL_WARP_CONSTANT && !(lo <= cval && cval <= hi);
    }
    const long long n = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int tfr = (int)(i / cols), tfc = (int)(i - (long long)tfr * cols);
        const double x = (double)tfc, y = (double)tfr;
        // _matrix_transform: each product and sum rounded (no contraction: -ffp-contract=off)
        double xx = M.m[0] * x;  xx = xx + M.m[1] * y;  xx = xx + M.m[2];
        double yy = M.m[3] * x;  yy = yy + M.m[4] * y;  yy = yy + M.m[5];
        double zz = M.m[6] * x;  zz = zz + M.m[7] * y;  zz = zz + M.m[8];
        const double c = xx / zz, r = yy / zz;
        const double fr = floor(r), fc = floor(c);
        // (a coordinate far outside any image: every neighbour is outside, clamp before the integer cast)
        const double big = 4.0e9;
        const long long minr = (long long)fmax(fmin(fr, big), -big), minc = (long long)fmax(fmin(fc, big), -big);
        const long long maxr = (long long)fmax(fmin(ceil(r), big), -big), maxc = (long long)fmax(fmin(ceil(c), big), -big);
        const double dr = r - fr, dc = c - fc;
        for (int ch = 0; ch < C; ++ch) {
            const double p00 = warp_pixel(im, h, w, C, ch, minr, minc, mode, cval);
            const double p01 = warp_pixel(im, h, w, C, ch, minr, maxc, mode, cval);
            const double p10 = warp_pixel(im, h, w, C, ch, maxr, minc, mode, cval);
            const double p11 = warp_pixel(im, h, w, C, ch, maxr, maxc, mode, cval);
            double top = (1.0 - dc) * p00;  top = top + dc * p01;
            double bot = (1.0 - dc) * p10;  bot = bot + dc * p11;
            double v = (1.0 - dr) * top;  v = v + dr * bot;
            if (clip && !(keep_cval && v == cval)) v = fmin(fmax(v, lo), hi);      // np.clip; cval pixels kept
            out[(size_t)i * C + ch] = v;
        }
    }
}

__global__ __launch_bounds__(256) void flip_boxes_kernel(float *__restrict__ boxes, int n, int stride, float width) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float *b = boxes + (size_t)i * stride;
    const float x1 = b[0], x2 = b[2];
    float a = width - x2;  a = a - 1.0f;
    float c = width - x1;  c = c - 1.0f;
    b[0] = a;
    b[2] = c;
}

}  // namespace wssdl

using namespace wssdl;

extern "C" size_t wssdl_image_prep_workspace_bytes(void) { return IMG_PARTS * sizeof(double); }

extern "C" int wssdl_image_prep(const uint8_t *gray, int h, int w, int row_stride, int flipped,
                                int use_brightness, float brightness_delta, int use_contrast,
                                float contrast_factor, double pixel_mean, float *out, void *workspace,
                                size_t workspace_bytes, wssdl_stream_t stream) {
    if (h < 1 || w < 1 || row_stride < w) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!gray || !out) return WSSDL_ERR_INVALID_ARGUMENT;
    if (use_contrast && (!workspace || workspace_bytes < IMG_PARTS * sizeof(double))) return WSSDL_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    double *parts = static_cast<double *>(workspace);
    if (use_contrast)
        hipLaunchKernelGGL(image_sum_kernel, dim3(IMG_PARTS), dim3(256), 0, st, gray, h, w, row_stride,
                           use_brightness, brightness_delta, parts);
    const long long n = (long long)h * w;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(image_prep_kernel, dim3(blocks), dim3(256), 0, st, gray, h, w, row_stride, flipped,
                       use_brightness, brightness_delta, use_contrast, contrast_factor, pixel_mean / 255.0, parts, out);
    return check_launch();
}

extern "C" int wssdl_image_adjust_f64(const double *im, int h, int w, int channels, int64_t row_stride,
                                      int use_brightness, double brightness_delta, int use_contrast,
                                      double contrast_factor, double pixel_mean, double *out, void *workspace,
                                      size_t workspace_bytes, wssdl_stream_t stream) {
    if (h < 1 || w < 1 || channels < 1 || row_stride < (int64_t)w * channels) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!im || !out) return WSSDL_ERR_INVALID_ARGUMENT;
    if (use_contrast && (!workspace || workspace_bytes < IMG_PARTS * sizeof(double))) return WSSDL_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    double *parts = static_cast<double *>(workspace);
    const int wc = w * channels;
    if (use_contrast)
        hipLaunchKernelGGL(image_sum_f64_kernel, dim3(IMG_PARTS), dim3(256), 0, st, im, h, wc, (long long)row_stride,
                           use_brightness, brightness_delta, parts);
    const long long n = (long long)h * wc;
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(image_adjust_f64_kernel, dim3(blocks), dim3(256), 0, st, im, h, wc, (long long)row_stride,
                       use_brightness, brightness_delta, use_contrast, contrast_factor, pixel_mean / 255.0, parts, out);
    return check_launch();
}

extern "C" int wssdl_image_to_blob(const void *im, int im_is_f64, int h, int w, double scale, int divide,
                                   float *blob, int index, int n_images, int Hmax, int Wmax,
                                   wssdl_stream_t stream) {
    if (h < 1 || w < 1 || Hmax < h || Wmax < w || index < 0 || index >= n_images) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!im || !blob || scale == 0.0) return WSSDL_ERR_INVALID_ARGUMENT;
    float *dst = blob + (size_t)index * Hmax * Wmax * 3;
    const long long n = (long long)Hmax * Wmax * 3;
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    if (im_is_f64)
        hipLaunchKernelGGL(image_to_blob_kernel<double>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           static_cast<const double *>(im), h, w, scale, divide, dst, Hmax, Wmax);
    else
        hipLaunchKernelGGL(image_to_blob_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream),
                           static_cast<const float *>(im), h, w, scale, divide, dst, Hmax, Wmax);
    return check_launch();
}

extern "C" size_t wssdl_image_warp_workspace_bytes(void) { return 2 * WARP_PARTS * sizeof(double); }

template <typename T>
static int launch_warp(const T *im, int h, int w, int C, const WarpMatrix &M, int rows, int cols, int mode, double cval,
                       int clip, double *out, double *parts, hipStream_t st) {
    if (clip)
        hipLaunchKernelGGL(image_minmax_kernel<T>, dim3(WARP_PARTS), dim3(256), 0, st, im, (long long)h * w * C, parts);
    const long long n = (long long)rows * cols;
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(image_warp_kernel<T>, dim3(blocks), dim3(256), 0, st, im, h, w, C, M, rows, cols, mode, cval, clip,
                       parts, out);
    return check_launch();
}

extern "C" int wssdl_image_warp(const void *im, int im_is_f64, int h, int w, int channels, const double *matrix,
                                int rows, int cols, int mode, double cval, int clip, double *out, void *workspace,
                                size_t workspace_bytes, wssdl_stream_t stream) {
    if (h < 1 || w < 1 || channels < 1 || rows < 1 || cols < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    if (!im || !matrix || !out || (mode != WSSDL_WARP_CONSTANT && mode != WSSDL_WARP_EDGE)) return WSSDL_ERR_INVALID_ARGUMENT;
    if (clip && (!workspace || workspace_bytes < wssdl_image_warp_workspace_bytes())) return WSSDL_ERR_WORKSPACE;
    WarpMatrix M;
    for (int i = 0; i < 9; ++i) M.m[i] = matrix[i];
    double *parts = static_cast<double *>(workspace);
    if (im_is_f64)
        return launch_warp(static_cast<const double *>(im), h, w, channels, M, rows, cols, mode, cval, clip, out, parts,
                           as_stream(stream));
    return launch_warp(static_cast<const float *>(im), h, w, channels, M, rows, cols, mode, cval, clip, out, parts,
                       as_stream(stream));
}

extern "C" int wssdl_image_resize(const void *im, int im_is_f64, int h, int w, int channels, int rows, int cols,
                                  double *out, void *workspace, size_t workspace_bytes, wssdl_stream_t stream) {
    if (h < 1 || w < 1 || rows < 1 || cols < 1) return WSSDL_ERR_INVALID_ARGUMENT;
    // resize(): output (col, row) -> input (col_scale * (col + 0.5) - 0.5, row_scale * (row + 0.5) - 0.5); a
    // translation to the input's centre for a 1 x 1 output (_warps.py `resize`, 0.14.2)
    double m[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
    if (rows == 1 && cols == 1) {
        m[2] = (double)w / 2.0 - 0.5;
        m[5] = (double)h / 2.0 - 0.5;
    } else {
        const double rs = (double)h / (double)rows, cs = (double)w / (double)cols;
        m[0] = cs;  m[2] = cs * 0.5 - 0.5;
        m[4] = rs;  m[5] = rs * 0.5 - 0.5;
    }
    return wssdl_image_warp(im, im_is_f64, h, w, channels, m, rows, cols, WSSDL_WARP_CONSTANT, 0.0, 1, out, workspace,
                            workspace_bytes, stream);
}

extern "C" int wssdl_flip_boxes(float *boxes, int n, int stride, float width, wssdl_stream_t stream) {
    if (n < 0 || stride < 4) return WSSDL_ERR_INVALID_ARGUMENT;
    if (n == 0) return WSSDL_OK;
    if (!boxes) return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(flip_boxes_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), boxes, n, stride, width);
    return check_launch();
}
