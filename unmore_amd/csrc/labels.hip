// Ground-truth synthesis on the device: the label half of the reference's dataset item
// (datasets.py:158-159,171-222, branch without random crop) for a whole batch of masks that are already at the
// training resolution:
//   sdf          = DT(mask)/max - DT(1-mask)/max            cv2.distanceTransform(u8, cv2.DIST_L2, 3)   (:176-190)
//   center_field = normalize(mask * normalize((i, j) - (c_y, c_x)))                                       (:193-207)
//   saliency     = mask > 0                                                                               (:212)
// An empty mask yields all-zero labels (the reference's early return, :128-138).
// The random-crop branch (:144-190, the one __getitem__ takes) adds: batched crop + resize (bilinear f32 / nearest u8),
// the distance transform on its own, and label synthesis around a foreground field computed before the crop.
//
// cv2.distanceTransform with DIST_L2 and a 3x3 mask is OpenCV's two-pass chamfer transform in 16.16 fixed point
// (imgproc/src/distransform.cpp, distanceTransform_3x3; a = 0.955, b = 1.3693 as documented for DIST_L2 3x3):
//   forward  (rows top-down, left-to-right):   t = src ? min(up-left + b, up + a, up-right + b, left + a) : 0
//   backward (rows bottom-up, right-to-left):  t = min(t, down-right + b, down + a, down-left + b, right + a)
//   out = min(t, INT_MAX >> 2) * 2^-16, with a one-pixel border initialised to INT_MAX >> 2.
// The left/right terms make a row sequential; they are min-plus prefix scans: with u[k] = c[k] - k*a the row is
// t[j] = prefixmin(u)[j] + j*a (exact in integers).  One workgroup per (image, polarity) walks the rows; a row is a
// block-wide prefix-min over per-thread chunks.
#include "umr_common.h"
#include <limits.h>

namespace {

constexpr int HV = 62587;      // cvRound(0.955  * 65536)
constexpr int DG = 89738;      // cvRound(1.3693 * 65536)
constexpr int INIT0 = INT_MAX >> 2;
constexpr int LT = 256;        // threads per workgroup
constexpr int MAXW = 4096;     // widest supported row

// block-wide inclusive prefix-min of vals[0..n) held as per-thread chunks [lo, hi) in LDS array a[]; forward or backward
__device__ __forceinline__ void block_scan_min(int* a, int* part, int lo, int hi, bool backward) {
    const int t = threadIdx.x;
    int m = INT_MAX;
    if (!backward) { for (int k = lo; k < hi; ++k) { m = min(m, a[k]); a[k] = m; } }
    else { for (int k = hi - 1; k >= lo; --k) { m = min(m, a[k]); a[k] = m; } }
    part[t] = m;
    __syncthreads();
    // exclusive prefix over the thread totals (Hillis-Steele on 256 entries, direction-aware)
    int carry = INT_MAX;
    if (!backward) { for (int k = 0; k < t; ++k) carry = min(carry, part[k]); }
    else { for (int k = LT - 1; k > t; --k) carry = min(carry, part[k]); }
    for (int k = lo; k < hi; ++k) a[k] = min(a[k], carry);
    __syncthreads();
}

// grid = npol*B: workgroup (b, pol).  pol 0: distance to the nearest 0 of `mask`; pol 1: of `1 - mask`; pol0 shifts the
// polarity index (npol == 1, pol0 == 1: background transform only).
__global__ __launch_bounds__(LT) void dt3x3_kernel(const uint8_t* __restrict__ mask, int* __restrict__ tmp, int* __restrict__ tmax, int H, int W,
                                                   int npol, int pol0) {
    __shared__ int prev[MAXW + 2];   // previous row with the one-pixel border (prev[j+1] = row value at column j)
    __shared__ int cur[MAXW];
    __shared__ int part[LT];
    const int b = blockIdx.x / npol, pol = pol0 + (int)(blockIdx.x - b * npol);
    const uint8_t* src = mask + (int64_t)b * H * W;
    int* out = tmp + (int64_t)blockIdx.x * H * W;
    const int t = threadIdx.x;
    const int chunk = (W + LT - 1) / LT;
    const int lo = min(W, t * chunk), hi = min(W, lo + chunk);
    for (int j = t; j < W + 2; j += LT) prev[j] = INIT0;
    __syncthreads();
    // ---- forward
    for (int i = 0; i < H; ++i) {
        for (int j = lo; j < hi; ++j) {
            const bool on = ((src[(int64_t)i * W + j] != 0) != (pol != 0));
            const int c = on ? min(min(prev[j] + DG, prev[j + 1] + HV), prev[j + 2] + DG) : 0;
            // left neighbour of column 0 is the border: candidate INIT0 + HV
            cur[j] = (j == 0 && on ? min(c, INIT0 + HV) : c) - j * HV;
        }
        __syncthreads();
        block_scan_min(cur, part, lo, hi, false);
        for (int j = lo; j < hi; ++j) {
            const int v = cur[j] + j * HV;
            prev[j + 1] = v;
            out[(int64_t)i * W + j] = v;
        }
        __syncthreads();
    }
    // ---- backward
    for (int j = t; j < W + 2; j += LT) prev[j] = INIT0;
    __syncthreads();
    int mx = 0;
    for (int i = H - 1; i >= 0; --i) {
        for (int j = lo; j < hi; ++j) {
            const int t0 = out[(int64_t)i * W + j];
            int c = min(t0, min(min(prev[j + 2] + DG, prev[j + 1] + HV), prev[j] + DG));
            if (j == W - 1) c = min(c, INIT0 + HV);
            cur[j] = c + j * HV;   // suffix scan: t[j] = min_k>=j (c[k] + (k - j) a) = suffixmin(c[k] + k a)[j] - j a
        }
        __syncthreads();
        block_scan_min(cur, part, lo, hi, true);
        for (int j = lo; j < hi; ++j) {
            int v = cur[j] - j * HV;
            prev[j + 1] = v;
            v = min(v, INIT0);
            out[(int64_t)i * W + j] = v;
            mx = max(mx, v);
        }
        __syncthreads();
    }
    part[t] = mx;
    __syncthreads();
    if (t == 0) {
        int m = 0;
        for (int k = 0; k < LT; ++k) m = max(m, part[k]);
        tmax[blockIdx.x] = m;
    }
}

// bounding-box centre of each mask: ((min x + max x)/2, (min y + max y)/2); count of foreground pixels
__global__ __launch_bounds__(LT) void bbox_center_kernel(const uint8_t* __restrict__ mask, float* __restrict__ center, int* __restrict__ count, int H, int W) {
    __shared__ int s[5][LT];
    const int b = blockIdx.x, t = threadIdx.x;
    const uint8_t* src = mask + (int64_t)b * H * W;
    int x0 = INT_MAX, x1 = -1, y0 = INT_MAX, y1 = -1, n = 0;
    for (int idx = t; idx < H * W; idx += LT) {
        if (src[idx]) { const int y = idx / W, x = idx - y * W; x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y); ++n; }
    }
    s[0][t] = x0; s[1][t] = x1; s[2][t] = y0; s[3][t] = y1; s[4][t] = n;
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < LT; ++k) { x0 = min(x0, s[0][k]); x1 = max(x1, s[1][k]); y0 = min(y0, s[2][k]); y1 = max(y1, s[3][k]); n += s[4][k]; }
        count[b] = n;
        if (center) { center[2 * b] = n ? (float)(x0 + x1) / 2.f : 0.f; center[2 * b + 1] = n ? (float)(y0 + y1) / 2.f : 0.f; }
    }
}

// fg_sdf_in != nullptr: the random-crop branch (datasets.py:161-190) -- the foreground field was computed before the crop and
// arrives resized; only the background transform (slot 0 of tmp / tmax then) is taken from this mask, and an empty mask is
// NOT short-circuited (the reference's emptiness test happens before the crop, :146-157).
__global__ void labels_kernel(const uint8_t* __restrict__ mask, const float* __restrict__ center, const int* __restrict__ count,
                              const int* __restrict__ tmp, const int* __restrict__ tmax, const float* __restrict__ fg_sdf_in,
                              float* __restrict__ center_field, float* __restrict__ saliency, float* __restrict__ sdf, int B, int H, int W,
                              int use_bg_sdf) {
    const int64_t total = (int64_t)B * H * W;
    const float scale = 1.f / 65536.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / ((int64_t)H * W));
        const int r = (int)(idx - (int64_t)b * H * W);
        const int i = r / W, j = r - i * W;
        float* cf = center_field + (int64_t)b * 2 * H * W;
        if (!fg_sdf_in && count[b] == 0) { cf[r] = 0.f; cf[(int64_t)H * W + r] = 0.f; saliency[idx] = 0.f; sdf[idx] = 0.f; continue; }
        const bool on = mask[idx] != 0;
        // sdf (datasets.py:176-190)
        const int npol = fg_sdf_in ? 1 : 2;
        float v;
        if (fg_sdf_in) {
            v = fg_sdf_in[idx];
        } else {
            const float fmx = (float)tmax[2 * b] * scale;
            v = (float)tmp[((int64_t)2 * b) * H * W + r] * scale;
            if (fmx > 0.f) v = v / fmx;
        }
        if (use_bg_sdf) {
            const float bmx = (float)tmax[npol * b + npol - 1] * scale;
            float w = (float)tmp[((int64_t)npol * b + npol - 1) * H * W + r] * scale;
            if (bmx > 0.f) w = w / bmx;
            v = v + w * -1.f;
        }
        sdf[idx] = v;
        saliency[idx] = on ? 1.f : 0.f;
        // center field (datasets.py:193-207): channel 0 = row offset, channel 1 = column offset; F.normalize eps 1e-12, twice
        float d0 = (float)i - center[2 * b + 1], d1 = (float)j - center[2 * b];
        float nrm = fmaxf(sqrtf(d0 * d0 + d1 * d1), 1e-12f);
        d0 = d0 / nrm; d1 = d1 / nrm;
        d0 = on ? d0 : 0.f; d1 = on ? d1 : 0.f;
        nrm = fmaxf(sqrtf(d0 * d0 + d1 * d1), 1e-12f);
        cf[r] = d0 / nrm;
        cf[(int64_t)H * W + r] = d1 / nrm;
    }
}

// DT(mask) * 2^-16 (/ its maximum when normalize != 0 and the maximum is > 0): datasets.py:162-164
__global__ void dt_to_float_kernel(const int* __restrict__ tmp, const int* __restrict__ tmax, float* __restrict__ out, int B, int64_t HW, int normalize) {
    const int64_t total = (int64_t)B * HW;
    const float scale = 1.f / 65536.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / HW);
        float v = (float)tmp[idx] * scale;
        const float mx = (float)tmax[b] * scale;
        if (normalize && mx > 0.f) v = v / mx;
        out[idx] = v;
    }
}

// crop + resize of a batch, one box per item: src [B,C,H,W] -> dst [B,C,Ho,Wo].  Bilinear: torchvision tensor Resize without
// antialias == F.interpolate(mode="bilinear", align_corners=False) (datasets.py:99,103).  Nearest: F.interpolate(mode="nearest"),
// source index = min(floor(dst * in / out), in - 1) (datasets.py:100,104).
template <typename T, bool NEAREST>
__global__ void crop_resize_batch_kernel(const T* __restrict__ src, const int32_t* __restrict__ boxes, T* __restrict__ dst, int B, int C,
                                         int H, int W, int Ho, int Wo) {
    const int64_t total = (int64_t)B * C * Ho * Wo;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo);
        int64_t r = idx / Wo;
        const int oy = (int)(r % Ho); r /= Ho;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        const int x1 = boxes[n * 4 + 0], y1 = boxes[n * 4 + 1], x2 = boxes[n * 4 + 2], y2 = boxes[n * 4 + 3];
        const int hc = y2 - y1, wc = x2 - x1;
        T v = (T)0;
        if (hc > 0 && wc > 0) {
            const float sh = (float)hc / (float)Ho, sw = (float)wc / (float)Wo;
            const T* base = src + (((int64_t)n * C + c) * H + y1) * W + x1;
            if (NEAREST) {
                int iy = (int)floorf((float)oy * sh), ix = (int)floorf((float)ox * sw);
                iy = iy > hc - 1 ? hc - 1 : iy;
                ix = ix > wc - 1 ? wc - 1 : ix;
                v = base[(int64_t)iy * W + ix];
            } else {
                const float sy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
                int iy = (int)sy, ix = (int)sx;
                if (iy > hc - 1) iy = hc - 1;
                if (ix > wc - 1) ix = wc - 1;
                const int dy = iy < hc - 1 ? 1 : 0, dx = ix < wc - 1 ? 1 : 0;
                const float ly1 = fminf(fmaxf(sy - (float)iy, 0.f), 1.f), lx1 = fminf(fmaxf(sx - (float)ix, 0.f), 1.f);
                const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
                const T* q = base + (int64_t)iy * W + ix;
                const float v00 = (float)q[0], v01 = (float)q[dx], v10 = (float)q[(int64_t)dy * W], v11 = (float)q[(int64_t)dy * W + dx];
                v = (T)(ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11));
            }
        }
        dst[idx] = v;
    }
}

}  // namespace

extern "C" int umr_crop_resize_batch(const void* src, const int32_t* boxes, void* dst, int B, int C, int H, int W, int Ho, int Wo,
                                     int nearest_u8, umr_stream_t stream) {
    UMR_CHECK_ARG(src && boxes && dst && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "crop_resize_batch: bad arguments");
    const int64_t total = (int64_t)B * C * Ho * Wo;
    int64_t g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    hipStream_t s = (hipStream_t)stream;
    if (nearest_u8) hipLaunchKernelGGL((crop_resize_batch_kernel<uint8_t, true>), dim3((unsigned)g), dim3(256), 0, s, (const uint8_t*)src, boxes, (uint8_t*)dst, B, C, H, W, Ho, Wo);
    else hipLaunchKernelGGL((crop_resize_batch_kernel<float, false>), dim3((unsigned)g), dim3(256), 0, s, (const float*)src, boxes, (float*)dst, B, C, H, W, Ho, Wo);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int64_t umr_distance_transform_workspace(int B, int H, int W) { return ((int64_t)B * H * W + B) * 4; }

extern "C" int umr_distance_transform(const uint8_t* mask, float* out, void* workspace, int64_t workspace_bytes, int B, int H, int W,
                                      int normalize, umr_stream_t stream) {
    UMR_CHECK_ARG(mask && out && workspace, "distance_transform: null pointer");
    UMR_CHECK_ARG(B > 0 && H > 0 && W > 0 && W <= MAXW, "distance_transform: bad geometry (W <= 4096)");
    UMR_CHECK_ARG(workspace_bytes >= umr_distance_transform_workspace(B, H, W), "distance_transform: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int* tmp = (int*)workspace;
    int* tmax = tmp + (int64_t)B * H * W;
    hipLaunchKernelGGL(dt3x3_kernel, dim3(B), dim3(LT), 0, s, mask, tmp, tmax, H, W, 1, 0);
    UMR_LAUNCH_CHECK();
    const int64_t total = (int64_t)B * H * W;
    int64_t g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(dt_to_float_kernel, dim3((unsigned)g), dim3(256), 0, s, tmp, tmax, out, B, (int64_t)H * W, normalize);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int64_t umr_label_synthesis_workspace(int B, int H, int W) {
    return ((int64_t)2 * B * H * W + 2 * B + B) * 4 + (int64_t)2 * B * 4;
}

static int label_synthesis_impl(const uint8_t* mask, const float* center_xy, const float* fg_sdf_in, float* center_field, float* saliency,
                                float* sdf, void* workspace, int64_t workspace_bytes, int B, int H, int W, int use_bg_sdf, umr_stream_t stream) {
    UMR_CHECK_ARG(mask && center_field && saliency && sdf && workspace, "label_synthesis: null pointer");
    UMR_CHECK_ARG(B > 0 && H > 0 && W > 0 && W <= MAXW, "label_synthesis: bad geometry (W <= 4096)");
    UMR_CHECK_ARG(workspace_bytes >= umr_label_synthesis_workspace(B, H, W), "label_synthesis: workspace too small");
    UMR_CHECK_ARG(!fg_sdf_in || center_xy, "label_synthesis: the pre-computed foreground field comes with explicit object centres");
    hipStream_t s = (hipStream_t)stream;
    int* tmp = (int*)workspace;
    int* tmax = tmp + (int64_t)2 * B * H * W;
    int* count = tmax + 2 * B;
    float* cbuf = (float*)(count + B);
    hipLaunchKernelGGL(bbox_center_kernel, dim3(B), dim3(LT), 0, s, mask, center_xy ? nullptr : cbuf, count, H, W);
    UMR_LAUNCH_CHECK();
    if (fg_sdf_in) hipLaunchKernelGGL(dt3x3_kernel, dim3(B), dim3(LT), 0, s, mask, tmp, tmax, H, W, 1, 1);   // background transform only
    else hipLaunchKernelGGL(dt3x3_kernel, dim3(2 * B), dim3(LT), 0, s, mask, tmp, tmax, H, W, 2, 0);
    UMR_LAUNCH_CHECK();
    const int64_t total = (int64_t)B * H * W;
    int64_t g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(labels_kernel, dim3((unsigned)g), dim3(256), 0, s, mask, center_xy ? center_xy : cbuf, count, tmp, tmax, fg_sdf_in,
                       center_field, saliency, sdf, B, H, W, use_bg_sdf);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_label_synthesis(const uint8_t* mask, const float* center_xy, float* center_field, float* saliency, float* sdf,
                                   void* workspace, int64_t workspace_bytes, int B, int H, int W, int use_bg_sdf, umr_stream_t stream) {
    return label_synthesis_impl(mask, center_xy, nullptr, center_field, saliency, sdf, workspace, workspace_bytes, B, H, W, use_bg_sdf, stream);
}

extern "C" int umr_label_synthesis_cropped(const uint8_t* mask, const float* center_xy, const float* fg_sdf, float* center_field,
                                           float* saliency, float* sdf, void* workspace, int64_t workspace_bytes, int B, int H, int W,
                                           int use_bg_sdf, umr_stream_t stream) {
    UMR_CHECK_ARG(fg_sdf != nullptr, "label_synthesis_cropped: null foreground field");
    return label_synthesis_impl(mask, center_xy, fg_sdf, center_field, saliency, sdf, workspace, workspace_bytes, B, H, W, use_bg_sdf, stream);
}
