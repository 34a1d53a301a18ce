// Host image path on the device, the part that can be pinned, gfx950.
//
// Reference: code/lib/utils/blob.py:34-79 (prep_im_for_blob), :19-32 (im_list_to_blob),
// roi_data_layer/minibatch_bus.py:269-272 (one grey plane stacked three times, horizontal flip),
// datasets/imdb.py:106-121 (box mirroring of flipped images).  skimage.transform.resize
// (blob.py:74-77) sits in the middle of prep_im_for_blob and is NOT implemented here: the library
// is absent and its version unpinned, so no oracle can pin it.  The path is cut there:
//   wssdl_image_prep     u8 grey plane -> f32 [h,w,3]: flip, /255, brightness (+delta, clip),
//                        contrast ((x - mean) * f + mean, clip), - pixel_mean/255   (what the
//                        reference hands to the resize); one pass, 1 B read + 12 B written per pixel
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

extern "C" int wssdl_flip_boxes(float *boxes, int n, int stride, float width, wssdl_stream_t stream) {
    if (n < 0 || stride < 4) return WSSDL_ERR_INVALID_ARGUMENT;
    if (n == 0) return WSSDL_OK;
    if (!boxes) return WSSDL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(flip_boxes_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), boxes, n, stride, width);
    return check_launch();
}
