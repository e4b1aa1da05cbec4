// Ground-truth synthesis on the device: the label half of the reference's dataset item
// (datasets.py:158-159,171-222, branch without random crop) for a whole batch of masks that are already at the
// training resolution:
//   sdf          = DT(mask)/max - DT(1-mask)/max            cv2.distanceTransform(u8, cv2.DIST_L2, 3)   (:176-190)
//   center_field = normalize(mask * normalize((i, j) - (c_y, c_x)))                                       (:193-207)
//   saliency     = mask > 0                                                                               (:212)
// An empty mask yields all-zero labels (the reference's early return, :128-138).
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

// grid = 2*B: workgroup (b, pol).  pol 0: distance to the nearest 0 of `mask`; pol 1: of `1 - mask`.
__global__ __launch_bounds__(LT) void dt3x3_kernel(const uint8_t* __restrict__ mask, int* __restrict__ tmp, int* __restrict__ tmax, int H, int W) {
    __shared__ int prev[MAXW + 2];   // previous row with the one-pixel border (prev[j+1] = row value at column j)
    __shared__ int cur[MAXW];
    __shared__ int part[LT];
    const int b = blockIdx.x >> 1, pol = blockIdx.x & 1;
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

__global__ void labels_kernel(const uint8_t* __restrict__ mask, const float* __restrict__ center, const int* __restrict__ count,
                              const int* __restrict__ tmp, const int* __restrict__ tmax, float* __restrict__ center_field,
                              float* __restrict__ saliency, float* __restrict__ sdf, int B, int H, int W, int use_bg_sdf) {
    const int64_t total = (int64_t)B * H * W;
    const float scale = 1.f / 65536.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / ((int64_t)H * W));
        const int r = (int)(idx - (int64_t)b * H * W);
        const int i = r / W, j = r - i * W;
        float* cf = center_field + (int64_t)b * 2 * H * W;
        if (count[b] == 0) { cf[r] = 0.f; cf[(int64_t)H * W + r] = 0.f; saliency[idx] = 0.f; sdf[idx] = 0.f; continue; }
        const bool on = mask[idx] != 0;
        // sdf (datasets.py:176-190)
        const float fmx = (float)tmax[2 * b] * scale;
        float v = (float)tmp[((int64_t)2 * b) * H * W + r] * scale;
        if (fmx > 0.f) v = v / fmx;
        if (use_bg_sdf) {
            const float bmx = (float)tmax[2 * b + 1] * scale;
            float w = (float)tmp[((int64_t)2 * b + 1) * H * W + r] * scale;
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

}  // namespace

extern "C" int64_t umr_label_synthesis_workspace(int B, int H, int W) {
    return ((int64_t)2 * B * H * W + 2 * B + B) * 4 + (int64_t)2 * B * 4;
}

extern "C" int umr_label_synthesis(const uint8_t* mask, const float* center_xy, float* center_field, float* saliency, float* sdf,
                                   void* workspace, int64_t workspace_bytes, int B, int H, int W, int use_bg_sdf, umr_stream_t stream) {
    UMR_CHECK_ARG(mask && center_field && saliency && sdf && workspace, "label_synthesis: null pointer");
    UMR_CHECK_ARG(B > 0 && H > 0 && W > 0 && W <= MAXW, "label_synthesis: bad geometry (W <= 4096)");
    UMR_CHECK_ARG(workspace_bytes >= umr_label_synthesis_workspace(B, H, W), "label_synthesis: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int* tmp = (int*)workspace;
    int* tmax = tmp + (int64_t)2 * B * H * W;
    int* count = tmax + 2 * B;
    float* cbuf = (float*)(count + B);
    hipLaunchKernelGGL(bbox_center_kernel, dim3(B), dim3(LT), 0, s, mask, center_xy ? nullptr : cbuf, count, H, W);
    UMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dt3x3_kernel, dim3(2 * B), dim3(LT), 0, s, mask, tmp, tmax, H, W);
    UMR_LAUNCH_CHECK();
    const int64_t total = (int64_t)B * H * W;
    int64_t g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(labels_kernel, dim3((unsigned)g), dim3(256), 0, s, mask, center_xy ? center_xy : cbuf, count, tmp, tmax, center_field,
                       saliency, sdf, B, H, W, use_bg_sdf);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
