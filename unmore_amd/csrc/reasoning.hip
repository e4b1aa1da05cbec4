// Field post-processing around the ObjectnessNet calls of unMORE's object reasoning ("next" rows f1/f2 of
// SURVEY.md section 8f): proposal crop + bilinear resize, centre-field peak picking, boundary-field box deltas.
// Reference: object_reasoning.py:301-337,398-410 (crop/resize), :360-377,525-557 + utils/misc.py:10-20 (peaks),
// :139-174 (update_bbox_with_boundary_fields).  HBM/latency-bound integer / small-stencil work: one workgroup per
// crop or per field map, LDS for the masks, fixed-order reductions (bitwise reproducible).
#include "umr_common.h"

namespace {

// ---------------------------------------------------------------- crop + resize (torchvision tensor Resize, bilinear,
// no antialias == F.interpolate(mode="bilinear", align_corners=False))
__global__ void crop_resize_kernel(const float* __restrict__ img, const int32_t* __restrict__ boxes, float* __restrict__ out, int N,
                                   int H, int W, int S) {
    const int64_t total = (int64_t)N * 3 * S * S;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % S);
        int64_t r = idx / S;
        const int oy = (int)(r % S); r /= S;
        const int c = (int)(r % 3);
        const int n = (int)(r / 3);
        const int x1 = boxes[n * 4 + 0], y1 = boxes[n * 4 + 1], x2 = boxes[n * 4 + 2], y2 = boxes[n * 4 + 3];
        const int hc = y2 - y1, wc = x2 - x1;
        float v = 0.f;
        if (hc > 0 && wc > 0) {
            const float sh = (float)hc / (float)S, sw = (float)wc / (float)S;
            const float sy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
            int iy = (int)sy, ix = (int)sx;
            if (iy > hc - 1) iy = hc - 1;
            if (ix > wc - 1) ix = wc - 1;
            const int dy = iy < hc - 1 ? 1 : 0, dx = ix < wc - 1 ? 1 : 0;
            const float ly1 = fminf(fmaxf(sy - (float)iy, 0.f), 1.f), lx1 = fminf(fmaxf(sx - (float)ix, 0.f), 1.f);
            const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const float* p = img + ((int64_t)c * H + (y1 + iy)) * W + (x1 + ix);
            const float v00 = p[0], v01 = p[dx], v10 = p[(int64_t)dy * W], v11 = p[(int64_t)dy * W + dx];
            v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        }
        out[idx] = v;
    }
}

// ---------------------------------------------------------------- centre peaks
// one workgroup per map; masks live in LDS as bytes (two planes for the erosion rounds)
struct PeakCfg { int H, W, border, erode_k, erode_rounds; };

__global__ __launch_bounds__(256) void center_peaks_kernel(const float* __restrict__ sdf, const float* __restrict__ center,
                                                           const double* __restrict__ filt /* [2][5][5] */, double* __restrict__ score_out,
                                                           double* __restrict__ maxval, int64_t* __restrict__ argmax, PeakCfg cfg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ double red_v[256];
    __shared__ int red_i[256];
    const int H = cfg.H, W = cfg.W, HW = H * W;
    unsigned char* m0 = lds;
    unsigned char* m1 = lds + HW;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = sdf + (int64_t)b * HW;
    const float* c0 = center + (int64_t)b * 2 * HW;
    const float* c1 = c0 + HW;
    // union of (sigmoid(sdf) > 0.5) and (||center|| > 0.5)  (object_reasoning.py:528-531)
    for (int i = tid; i < HW; i += 256) {
        const float sg = 1.0f / (1.0f + expf(-s[i]));
        const float nr = sqrtf(c0[i] * c0[i] + c1[i] * c1[i]);
        m0[i] = (sg > 0.5f || nr > 0.5f) ? 1 : 0;
    }
    __syncthreads();
    // erosion: k x k all-ones box, keep where the (zero padded) sum == k*k  (utils/misc.py:10-20) -- separable AND
    const int rad = (cfg.erode_k - 1) / 2;
    for (int round = 0; round < cfg.erode_rounds; ++round) {
        for (int i = tid; i < HW; i += 256) {  // horizontal
            const int y = i / W, x = i - y * W;
            unsigned char ok = 1;
            for (int d = -rad; d <= rad; ++d) {
                const int xx = x + d;
                ok &= (xx >= 0 && xx < W) ? m0[y * W + xx] : 0;
            }
            m1[i] = ok;
        }
        __syncthreads();
        for (int i = tid; i < HW; i += 256) {  // vertical
            const int y = i / W, x = i - y * W;
            unsigned char ok = 1;
            for (int d = -rad; d <= rad; ++d) {
                const int yy = y + d;
                ok &= (yy >= 0 && yy < H) ? m1[yy * W + x] : 0;
            }
            m0[i] = ok;
        }
        __syncthreads();
    }
    // anti-centre score: float64 5x5 correlation with normalize((2-i, 2-j)), /24  (object_reasoning.py:360-377)
    double best = -1.0e300;
    int besti = 0;
    bool any = false;
    for (int i = tid; i < HW; i += 256) {
        const int y = i / W, x = i - y * W;
        double acc = 0.0;
        if (m0[i] && y >= cfg.border && y < H - cfg.border && x >= cfg.border && x < W - cfg.border) {
            for (int ch = 0; ch < 2; ++ch) {
                const float* cp = ch == 0 ? c0 : c1;
                for (int fi = 0; fi < 5; ++fi) {
                    const int yy = y + fi - 2;
                    if (yy < 0 || yy >= H) continue;
                    for (int fj = 0; fj < 5; ++fj) {
                        const int xx = x + fj - 2;
                        if (xx < 0 || xx >= W) continue;
                        acc += (double)cp[yy * W + xx] * filt[(ch * 5 + fi) * 5 + fj];
                    }
                }
            }
            acc = acc / 24.0;
        }
        if (score_out) score_out[(int64_t)b * HW + i] = acc;
        if (!any || acc > best) { best = acc; besti = i; any = true; }  // i increases: first maximum wins inside a thread
    }
    red_v[tid] = any ? best : -1.0e300;
    red_i[tid] = any ? besti : 0x7FFFFFFF;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            const double v2 = red_v[tid + off];
            const int i2 = red_i[tid + off];
            if (v2 > red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) { maxval[b] = red_v[0]; argmax[b] = red_i[0]; }
}

// ---------------------------------------------------------------- centre peaks with an argmax certificate
// The same peak picking, plus: is the flat argmax PROVABLY the one any pair of fields within `eps` (max-norm) of these would give?
// (the argument of oracle/objectness_oracle.py::peak_certificate, restated on the device so that a sweep can run in a cheaper
// arithmetic mode and re-run only what it cannot certify -- object_reasoning.py:525-557 is what consumes the index.)
// Erosion is monotone: with F = the pixels whose union-mask decision (object_reasoning.py:528-533) a perturbation < eps can flip,
// every reachable eroded mask lies between erode(union & ~F) and erode(union | F); a score moves by at most sqrt(2) * eps (24
// unit-vector taps / 24).  Certified when
//   (positive peak)  amax > 0, the peak survives in the SMALLEST mask, beats every other pixel of the LARGEST mask (and the zeros
//                    outside it) by more than 2 sqrt(2) eps, and amax stays on its side of the singularity threshold (:541);
//   (no peak)        amax == 0 at index 0 (a border pixel: exactly zero under any perturbation) and every pixel of the largest mask
//                    scores below -2 sqrt(2) eps (or that mask is empty inside the border): the map stays without a positive score.
// Six byte planes of H*W in LDS (three masks x two erosion buffers).
__global__ __launch_bounds__(256) void center_peaks_cert_kernel(const float* __restrict__ sdf, const float* __restrict__ center,
                                                                const double* __restrict__ filt, double* __restrict__ maxval,
                                                                int64_t* __restrict__ argmax, int32_t* __restrict__ certified, PeakCfg cfg,
                                                                float eps, double thres) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ double red_v[256], red_a[256], red_b[256];
    __shared__ int red_i[256], red_ai[256];
    const int H = cfg.H, W = cfg.W, HW = H * W;
    unsigned char* mk[3][2];
    for (int k = 0; k < 3; ++k) { mk[k][0] = lds + (2 * k) * HW; mk[k][1] = lds + (2 * k + 1) * HW; }
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = sdf + (int64_t)b * HW;
    const float* c0 = center + (int64_t)b * 2 * HW;
    const float* c1 = c0 + HW;
    for (int i = tid; i < HW; i += 256) {
        const float sv = s[i];
        const float sg = 1.0f / (1.0f + expf(-sv));
        const float nr = sqrtf(c0[i] * c0[i] + c1[i] * c1[i]);
        const bool sb = sg > 0.5f, cb = nr > 0.5f;
        const unsigned char u = (sb || cb) ? 1 : 0;
        // distance of this pixel's decision to its threshold, in field units: |sdf| (sigmoid(s) > 0.5 <=> s > 0), | ||c|| - 0.5 |
        const float ds = fabsf(sv), dc = fabsf(nr - 0.5f);
        const float flip = (sb && cb) ? fmaxf(ds, dc) : sb ? ds : cb ? dc : fminf(ds, dc);
        const bool f = flip < eps;
        mk[0][0][i] = u;
        mk[1][0][i] = f ? 0 : u;
        mk[2][0][i] = f ? 1 : u;
    }
    __syncthreads();
    const int rad = (cfg.erode_k - 1) / 2;
    for (int round = 0; round < cfg.erode_rounds; ++round) {
        for (int i = tid; i < HW; i += 256) {  // horizontal
            const int y = i / W, x = i - y * W;
            unsigned char ok0 = 1, ok1 = 1, ok2 = 1;
            for (int d = -rad; d <= rad; ++d) {
                const int xx = x + d;
                const bool in = xx >= 0 && xx < W;
                ok0 &= in ? mk[0][0][y * W + xx] : 0;
                ok1 &= in ? mk[1][0][y * W + xx] : 0;
                ok2 &= in ? mk[2][0][y * W + xx] : 0;
            }
            mk[0][1][i] = ok0; mk[1][1][i] = ok1; mk[2][1][i] = ok2;
        }
        __syncthreads();
        for (int i = tid; i < HW; i += 256) {  // vertical
            const int y = i / W, x = i - y * W;
            unsigned char ok0 = 1, ok1 = 1, ok2 = 1;
            for (int d = -rad; d <= rad; ++d) {
                const int yy = y + d;
                const bool in = yy >= 0 && yy < H;
                ok0 &= in ? mk[0][1][yy * W + x] : 0;
                ok1 &= in ? mk[1][1][yy * W + x] : 0;
                ok2 &= in ? mk[2][1][yy * W + x] : 0;
            }
            mk[0][0][i] = ok0; mk[1][0][i] = ok1; mk[2][0][i] = ok2;
        }
        __syncthreads();
    }
    // scores on the largest mask (a superset of the actual one).  Per thread: the actual map's first maximum (as center_peaks_kernel),
    // and the two largest scores of the largest mask with the first one's index
    double best = -1.0e300, top1 = -1.0e300, top2 = -1.0e300;
    int besti = 0, top1i = -1;
    bool any = false;
    for (int i = tid; i < HW; i += 256) {
        const int y = i / W, x = i - y * W;
        const bool inb = y >= cfg.border && y < H - cfg.border && x >= cfg.border && x < W - cfg.border;
        double acc = 0.0;
        const bool in_max = mk[2][0][i] && inb;
        if (in_max) {
            for (int ch = 0; ch < 2; ++ch) {
                const float* cp = ch == 0 ? c0 : c1;
                for (int fi = 0; fi < 5; ++fi) {
                    const int yy = y + fi - 2;
                    if (yy < 0 || yy >= H) continue;
                    for (int fj = 0; fj < 5; ++fj) {
                        const int xx = x + fj - 2;
                        if (xx < 0 || xx >= W) continue;
                        acc += (double)cp[yy * W + xx] * filt[(ch * 5 + fi) * 5 + fj];
                    }
                }
            }
            acc = acc / 24.0;
            if (acc > top1) { top2 = top1; top1 = acc; top1i = i; } else if (acc > top2) { top2 = acc; }
        }
        const double actual = mk[0][0][i] ? acc : 0.0;      // (the actual mask is inside the largest one)
        if (!any || actual > best) { best = actual; besti = i; any = true; }
    }
    red_v[tid] = any ? best : -1.0e300;
    red_i[tid] = any ? besti : 0x7FFFFFFF;
    red_a[tid] = top1; red_ai[tid] = top1i; red_b[tid] = top2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            const double v2 = red_v[tid + off];
            const int i2 = red_i[tid + off];
            if (v2 > red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
            // merge two (top1, top2) pairs
            const double a1 = red_a[tid], b1 = red_b[tid], a2 = red_a[tid + off], b2 = red_b[tid + off];
            if (a2 > a1) { red_a[tid] = a2; red_ai[tid] = red_ai[tid + off]; red_b[tid] = a1 > b2 ? a1 : b2; }
            else         { red_b[tid] = a2 > b1 ? a2 : b1; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double amax = red_v[0];
        const int p = red_i[0];
        maxval[b] = amax;
        argmax[b] = p;
        const double bound = 2.0 * 1.4142135623730951 * (double)eps;
        int cert = 0;
        if (amax > 0.0) {
            // every other pixel of the largest mask, and the exact zeros outside it
            double runner = red_ai[0] == p ? red_b[0] : red_a[0];
            if (runner < 0.0 && HW > 1) runner = 0.0;
            cert = mk[1][0][p] && (amax - runner > bound) && (fabs(amax - thres) > bound);
        } else {
            cert = amax == 0.0 && p == 0 && cfg.border >= 1 && (red_ai[0] < 0 || red_a[0] < -bound) && (fabs(thres) > bound);
        }
        certified[b] = cert;
    }
}

// ---------------------------------------------------------------- boundary-field box deltas (object_reasoning.py:139-174)
__global__ __launch_bounds__(256) void boundary_deltas_kernel(const float* __restrict__ sdf, float* __restrict__ deltas, int H, int W) {
    __shared__ float red[4][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = sdf + (int64_t)b * H * W;
    const int h1 = H - 1, w1 = W - 1;
    float s_fg = 0.f, s_bg = 0.f, n_fg = 0.f, n_bg = 0.f;
    for (int i = tid; i < h1 * w1; i += 256) {
        const int y = i / w1, x = i - y * w1;
        const float v = s[y * W + x];
        const float dy = s[(y + 1) * W + x] - v, dx = s[y * W + x + 1] - v;
        const float nr = sqrtf(dy * dy + dx * dx);
        const float fg = 1.0f / (1.0f + expf(-v)), bg = 1.0f - fg;
        s_fg += fg * nr; s_bg += bg * nr; n_fg += fg; n_bg += bg;
    }
    red[0][tid] = s_fg; red[1][tid] = s_bg; red[2][tid] = n_fg; red[3][tid] = n_bg;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + off];
        __syncthreads();
    }
    const float avg_fg = red[0][0] / (red[2][0] + 1e-8f), avg_bg = red[1][0] / (red[3][0] + 1e-8f);
    const float step_fg = 1.0f / (avg_fg + 1e-10f), step_bg = 1.0f / (avg_bg + 1e-10f);
    __syncthreads();
    auto movement = [&](int y, int x) {
        const float v = s[y * W + x];
        const float fg = 1.0f / (1.0f + expf(-v));
        return (step_fg * fg + step_bg * (1.0f - fg)) * v;
    };
    float mx1 = -INFINITY, my1 = -INFINITY, mx2 = -INFINITY, my2 = -INFINITY;
    for (int y = tid; y < h1; y += 256) { mx1 = fmaxf(mx1, movement(y, 0)); mx2 = fmaxf(mx2, movement(y, w1 - 1)); }
    for (int x = tid; x < w1; x += 256) { my1 = fmaxf(my1, movement(0, x)); my2 = fmaxf(my2, movement(h1 - 1, x)); }
    red[0][tid] = mx1; red[1][tid] = my1; red[2][tid] = mx2; red[3][tid] = my2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) for (int k = 0; k < 4; ++k) red[k][tid] = fmaxf(red[k][tid], red[k][tid + off]);
        __syncthreads();
    }
    if (tid == 0) {
        deltas[b * 4 + 0] = -red[0][0];
        deltas[b * 4 + 1] = -red[1][0];
        deltas[b * 4 + 2] = red[2][0];
        deltas[b * 4 + 3] = red[3][0];
    }
}


// ---- greedy non-maximum suppression (object_reasoning.py:661: torchvision.ops.nms(boxes, scores, iou_threshold)) -------------
// Rank r = position in `order` (the caller's descending-score order).  Pass 1: one bit per (rank i, rank j > i) -- does box i
// suppress box j, i.e. IoU(i, j) > threshold, in torchvision's own f32 formula (inter / (area_i + area_j - inter)).  Pass 2: one
// wave walks the ranks in order; lane l holds word l of the "removed" bit set (n <= 4096 boxes), a rank that is not removed is kept
// and ORs its row into the set.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ order, int n, float thr,
                                                      unsigned long long* __restrict__ mask, int words) {
    const int i = blockIdx.y, wj = blockIdx.x;
    const int j = wj * 64 + threadIdx.x;
    const float* a = boxes + order[i] * 4;
    const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    bool sup = false;
    if (j < n && j > i) {
        const float* b = boxes + order[j] * 4;
        const float left = fmaxf(a0, b[0]), right = fminf(a2, b[2]);
        const float top = fmaxf(a1, b[1]), bottom = fminf(a3, b[3]);
        const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
        const float inter = w * h;
        const float sa = (a2 - a0) * (a3 - a1), sb = (b[2] - b[0]) * (b[3] - b[1]);
        sup = (inter / (sa + sb - inter)) > thr;
    }
    const unsigned long long bits = __ballot(sup);
    if (threadIdx.x == 0) mask[(int64_t)i * words + wj] = bits;
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int64_t* __restrict__ order, int n, int words,
                                                      int64_t* __restrict__ keep, int32_t* __restrict__ n_keep) {
    const int lane = threadIdx.x;
    unsigned long long removed = 0ull;      // word `lane` of the set
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long wi = __shfl(removed, i >> 6);
        if (!((wi >> (i & 63)) & 1ull)) {   // wave-uniform
            if (lane == 0) keep[kept] = order[i];
            ++kept;
            if (lane < words) removed |= mask[(int64_t)i * words + lane];
        }
    }
    if (lane == 0) *n_keep = kept;
}


// ---- object scoring (object_scoring.py:172-272): the two binary masks of a proposal (||center|| > 0.5, sigmoid(sdf) > 0.5, on its
// S x S crop) are resized to the proposal's box with torchvision's tensor Resize -- for an integer tensor: bilinear in f32
// (align_corners=False), then torch.round (half to even: 1 iff the value exceeds 0.5) -- pasted into an image-sized canvas and OR-ed
// (:196-228).  Both masks live in LDS as bytes; the bilinear arithmetic is PyTorch's, unfused: h0 * (w0 * p00 + w1 * p01) + h1 * (...).
struct PasteAxis { int i0, i1; float l0, l1; };
__device__ __forceinline__ PasteAxis paste_axis(int o, int out_size, int S) {
    const float scale = (float)S / (float)out_size;
    const float src = fmaxf(__fsub_rn(__fmul_rn(scale, (float)o + 0.5f), 0.5f), 0.f);
    int i0 = (int)floorf(src);
    if (i0 > S - 1) i0 = S - 1;
    const float l1 = fminf(fmaxf(__fsub_rn(src, (float)i0), 0.f), 1.f);
    return PasteAxis{i0, i0 < S - 1 ? i0 + 1 : i0, __fsub_rn(1.f, l1), l1};
}
__device__ __forceinline__ bool paste_bit(const unsigned char* m, int S, const PasteAxis& ay, const PasteAxis& ax) {
    const float p00 = (float)m[ay.i0 * S + ax.i0], p01 = (float)m[ay.i0 * S + ax.i1];
    const float p10 = (float)m[ay.i1 * S + ax.i0], p11 = (float)m[ay.i1 * S + ax.i1];
    const float top = __fadd_rn(__fmul_rn(ax.l0, p00), __fmul_rn(ax.l1, p01));
    const float bot = __fadd_rn(__fmul_rn(ax.l0, p10), __fmul_rn(ax.l1, p11));
    return __fadd_rn(__fmul_rn(ay.l0, top), __fmul_rn(ay.l1, bot)) > 0.5f;
}
// fills the two LDS mask planes of proposal b; returns (this thread's) partial maxima of ||center|| and sdf
__device__ __forceinline__ void paste_masks_to_lds(const float* __restrict__ sdf, const float* __restrict__ center, int b, int S, unsigned char* mc,
                                                   unsigned char* mb, float& max_norm, float& max_sdf) {
    const int SS = S * S;
    const float* s = sdf + (int64_t)b * SS;
    const float* c0 = center + (int64_t)b * 2 * SS;
    const float* c1 = c0 + SS;
    for (int i = threadIdx.x; i < SS; i += blockDim.x) {
        const float sv = s[i];
        const float sg = 1.0f / (1.0f + expf(-sv));
        const float nr = sqrtf(c0[i] * c0[i] + c1[i] * c1[i]);
        mc[i] = nr > 0.5f ? 1 : 0;
        mb[i] = sg > 0.5f ? 1 : 0;
        max_norm = fmaxf(max_norm, nr);
        max_sdf = fmaxf(max_sdf, sv);
    }
}

__global__ __launch_bounds__(256) void mask_paste_stats_kernel(const float* __restrict__ sdf, const float* __restrict__ center, const int32_t* __restrict__ boxes,
                                                               int S, int32_t* __restrict__ stats, float* __restrict__ maxima) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int r_xmin[256], r_ymin[256], r_xmax[256], r_ymax[256], r_area[256];
    __shared__ float r_mn[256], r_ms[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned char* mc = lds;
    unsigned char* mb = lds + S * S;
    float mn = -INFINITY, ms = -INFINITY;
    paste_masks_to_lds(sdf, center, b, S, mc, mb, mn, ms);
    __syncthreads();
    const int x1 = boxes[b * 4 + 0], y1 = boxes[b * 4 + 1], x2 = boxes[b * 4 + 2], y2 = boxes[b * 4 + 3];
    const int hb = y2 - y1, wb = x2 - x1;
    int xmin = 0x7FFFFFFF, ymin = 0x7FFFFFFF, xmax = -1, ymax = -1, area = 0;
    if (hb > 0 && wb > 0) {
        const int64_t total = (int64_t)hb * wb;
        for (int64_t i = tid; i < total; i += 256) {
            const int oy = (int)(i / wb), ox = (int)(i - (int64_t)oy * wb);
            const PasteAxis ay = paste_axis(oy, hb, S), ax = paste_axis(ox, wb, S);
            if (paste_bit(mc, S, ay, ax) || paste_bit(mb, S, ay, ax)) {
                const int x = x1 + ox, y = y1 + oy;
                xmin = min(xmin, x); xmax = max(xmax, x); ymin = min(ymin, y); ymax = max(ymax, y);
                ++area;
            }
        }
    }
    r_xmin[tid] = xmin; r_ymin[tid] = ymin; r_xmax[tid] = xmax; r_ymax[tid] = ymax; r_area[tid] = area; r_mn[tid] = mn; r_ms[tid] = ms;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            r_xmin[tid] = min(r_xmin[tid], r_xmin[tid + st]); r_ymin[tid] = min(r_ymin[tid], r_ymin[tid + st]);
            r_xmax[tid] = max(r_xmax[tid], r_xmax[tid + st]); r_ymax[tid] = max(r_ymax[tid], r_ymax[tid + st]);
            r_area[tid] += r_area[tid + st];
            r_mn[tid] = fmaxf(r_mn[tid], r_mn[tid + st]); r_ms[tid] = fmaxf(r_ms[tid], r_ms[tid + st]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const bool any = r_area[0] > 0;      // an empty mask has the box [0, 0, 0, 0] (pycocotools toBbox of an empty RLE)
        stats[b * 5 + 0] = any ? r_xmin[0] : 0; stats[b * 5 + 1] = any ? r_ymin[0] : 0;
        stats[b * 5 + 2] = any ? r_xmax[0] + 1 : 0; stats[b * 5 + 3] = any ? r_ymax[0] + 1 : 0;
        stats[b * 5 + 4] = r_area[0];
        maxima[b * 2 + 0] = r_mn[0]; maxima[b * 2 + 1] = r_ms[0];
    }
}

__global__ __launch_bounds__(256) void mask_paste_kernel(const float* __restrict__ sdf, const float* __restrict__ center, const int32_t* __restrict__ boxes,
                                                         const int64_t* __restrict__ select, int S, int H, int W, unsigned char* __restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int b = (int)select[k];
    unsigned char* mc = lds;
    unsigned char* mb = lds + S * S;
    float mn = 0.f, ms = 0.f;
    paste_masks_to_lds(sdf, center, b, S, mc, mb, mn, ms);
    __syncthreads();
    const int x1 = boxes[b * 4 + 0], y1 = boxes[b * 4 + 1], x2 = boxes[b * 4 + 2], y2 = boxes[b * 4 + 3];
    const int hb = y2 - y1, wb = x2 - x1;
    unsigned char* out = masks + (int64_t)k * H * W;
    for (int64_t i = tid; i < (int64_t)H * W; i += 256) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        unsigned char v = 0;
        if (hb > 0 && wb > 0 && x >= x1 && x < x2 && y >= y1 && y < y2) {
            const PasteAxis ay = paste_axis(y - y1, hb, S), ax = paste_axis(x - x1, wb, S);
            v = (paste_bit(mc, S, ay, ax) || paste_bit(mb, S, ay, ax)) ? 1 : 0;
        }
        out[i] = v;
    }
}


// ---- connected components of the union mask (object_reasoning.py:206-257 with --analyze_cc, README.md:176: scipy.ndimage.label with a
// 3 x 3 structure = 8-connectivity, find_objects) -- one workgroup per map, labels in LDS.  Label equivalence (Hawick et al.): every mask
// pixel starts as its own root (label = flat index + 1); a sweep lowers a pixel's label to the smallest label among its 8 neighbours and
// hands that label to its old root, a compression pass makes every pixel point at its root; repeat until a sweep changes nothing.  At the
// end a component's label is its first pixel in raster order -- scipy numbers components in exactly that order -- so rank = number of roots
// before it.  Boxes [x1, y1, x2, y2) of the first `maxc` components, and the true count.
__global__ __launch_bounds__(256) void mask_components_kernel(const float* __restrict__ sdf, const float* __restrict__ center, int S, int maxc,
                                                              int32_t* __restrict__ counts, int32_t* __restrict__ boxes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int changed, chunk_roots[256];
    const int SS = S * S, b = blockIdx.x, tid = threadIdx.x;
    int* Lb = (int*)lds;                 // label plane
    int* Rk = Lb + SS;                   // rank of a root pixel
    int* bx = Rk + SS;                   // [maxc][4]: xmin, ymin, xmax, ymax
    const float* s = sdf + (int64_t)b * SS;
    const float* c0 = center + (int64_t)b * 2 * SS;
    const float* c1 = c0 + SS;
    for (int i = tid; i < SS; i += 256) {
        const float sg = 1.0f / (1.0f + expf(-s[i]));
        const float nr = sqrtf(c0[i] * c0[i] + c1[i] * c1[i]);
        Lb[i] = (sg > 0.5f || nr > 0.5f) ? i + 1 : 0;
    }
    for (int i = tid; i < maxc * 4; i += 256) bx[i] = (i & 2) ? -1 : 0x7FFFFFFF;
    __syncthreads();
    for (int guard = 0; guard < 4 * SS; ++guard) {        // (terminates long before: every sweep that changes something lowers a label)
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int i = tid; i < SS; i += 256) {
            const int l = Lb[i];
            if (l == 0) continue;
            const int y = i / S, x = i - y * S;
            int m = l;
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= S) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= S) continue;
                    const int q = Lb[yy * S + xx];
                    if (q != 0 && q < m) m = q;
                }
            }
            if (m < l) {
                atomicMin(&Lb[l - 1], m);      // the old root learns of the smaller label
                atomicMin(&Lb[i], m);
                changed = 1;
            }
        }
        __syncthreads();
        for (int i = tid; i < SS; i += 256) {  // compression: follow the chain to a pixel that is its own root
            int l = Lb[i];
            if (l == 0) continue;
            int r = Lb[l - 1];
            while (r < l) { l = r; r = Lb[l - 1]; }
            Lb[i] = l;
        }
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    // ranks of the roots in raster order: thread t owns the contiguous chunk of pixels [t * per, (t + 1) * per)
    const int per = (SS + 255) / 256;
    const int lo = tid * per, hi = min(SS, lo + per);
    int n_mine = 0;
    for (int i = lo; i < hi; ++i) n_mine += (Lb[i] == i + 1);
    chunk_roots[tid] = n_mine;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int t = 0; t < 256; ++t) { const int v = chunk_roots[t]; chunk_roots[t] = acc; acc += v; }
        counts[b] = acc;
    }
    __syncthreads();
    int rank = chunk_roots[tid];
    for (int i = lo; i < hi; ++i) {
        if (Lb[i] == i + 1) Rk[i] = rank++;
    }
    __syncthreads();
    for (int i = tid; i < SS; i += 256) {
        const int l = Lb[i];
        if (l == 0) continue;
        const int r = Rk[l - 1];
        if (r < maxc) {
            const int y = i / S, x = i - y * S;
            atomicMin(&bx[r * 4 + 0], x); atomicMin(&bx[r * 4 + 1], y);
            atomicMax(&bx[r * 4 + 2], x); atomicMax(&bx[r * 4 + 3], y);
        }
    }
    __syncthreads();
    const int n = min(counts[b], maxc);
    for (int i = tid; i < maxc; i += 256) {
        int32_t* o = boxes + ((int64_t)b * maxc + i) * 4;
        if (i < n) { o[0] = bx[i * 4 + 0]; o[1] = bx[i * 4 + 1]; o[2] = bx[i * 4 + 2] + 1; o[3] = bx[i * 4 + 3] + 1; }
        else { o[0] = o[1] = o[2] = o[3] = 0; }
    }
}

}  // namespace

extern "C" int umr_crop_resize_bilinear(const float* image, const int32_t* boxes, float* out, int N, int H, int W, int S,
                                        umr_stream_t stream) {
    UMR_CHECK_ARG(image && boxes && out && N > 0 && H > 0 && W > 0 && S > 0, "crop_resize: bad arguments");
    const int64_t total = (int64_t)N * 3 * S * S;
    int64_t g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(crop_resize_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, image, boxes, out, N, H, W, S);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_center_peaks(const float* sdf_maps, const float* center_fields, const double* filter50, double* score_out,
                                double* max_values, int64_t* argmax, int B, int H, int W, int border, int erode_kernel,
                                int erode_rounds, umr_stream_t stream) {
    UMR_CHECK_ARG(sdf_maps && center_fields && filter50 && max_values && argmax, "center_peaks: null pointer");
    UMR_CHECK_ARG(B > 0 && H > 0 && W > 0 && border >= 0 && erode_kernel >= 1 && (erode_kernel & 1) && erode_rounds >= 0,
                  "center_peaks: bad arguments");
    if ((int64_t)H * W * 2 > 150 * 1024) return umr_set_error(UMR_ERR_UNSUPPORTED, "center_peaks: map larger than the LDS mask planes (H*W <= 76800)");
    PeakCfg cfg{H, W, border, erode_kernel, erode_rounds};
    const size_t lds = (size_t)H * W * 2;
    UMR_SET_MAX_LDS_ONCE(center_peaks_kernel, 150 * 1024);
    hipLaunchKernelGGL(center_peaks_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, sdf_maps, center_fields, filter50, score_out,
                       max_values, argmax, cfg);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_center_peaks_certified(const float* sdf_maps, const float* center_fields, const double* filter50, double* max_values,
                                          int64_t* argmax, int32_t* certified, int B, int H, int W, int border, int erode_kernel,
                                          int erode_rounds, float eps, double singular_threshold, umr_stream_t stream) {
    UMR_CHECK_ARG(sdf_maps && center_fields && filter50 && max_values && argmax && certified, "center_peaks_certified: null pointer");
    UMR_CHECK_ARG(B > 0 && H > 0 && W > 0 && border >= 0 && erode_kernel >= 1 && (erode_kernel & 1) && erode_rounds >= 0 && eps >= 0.f,
                  "center_peaks_certified: bad arguments");
    if ((int64_t)H * W * 6 > 150 * 1024) return umr_set_error(UMR_ERR_UNSUPPORTED, "center_peaks_certified: map larger than the LDS mask planes (H*W <= 25600)");
    PeakCfg cfg{H, W, border, erode_kernel, erode_rounds};
    const size_t lds = (size_t)H * W * 6;
    UMR_SET_MAX_LDS_ONCE(center_peaks_cert_kernel, 150 * 1024);
    hipLaunchKernelGGL(center_peaks_cert_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, sdf_maps, center_fields, filter50, max_values,
                       argmax, certified, cfg, eps, singular_threshold);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_boundary_deltas(const float* sdf_maps, float* deltas, int B, int H, int W, umr_stream_t stream) {
    UMR_CHECK_ARG(sdf_maps && deltas && B > 0 && H > 1 && W > 1, "boundary_deltas: bad arguments");
    hipLaunchKernelGGL(boundary_deltas_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, sdf_maps, deltas, H, W);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int64_t umr_nms_workspace(int n) { return n > 0 ? (int64_t)n * ((n + 63) / 64) * 8 : 0; }

extern "C" int umr_nms(const float* boxes, const int64_t* order, int n, float iou_threshold, void* workspace, int64_t workspace_bytes,
                       int64_t* keep, int32_t* n_keep, umr_stream_t stream) {
    UMR_CHECK_ARG(boxes && order && keep && n_keep && n > 0, "nms: bad arguments");
    if (n > 4096) return umr_set_error(UMR_ERR_UNSUPPORTED, "nms: at most 4096 boxes (one wave holds the removed set)");
    UMR_CHECK_ARG(workspace && workspace_bytes >= umr_nms_workspace(n) && ((uintptr_t)workspace & 7) == 0, "nms: workspace smaller than umr_nms_workspace(n) or misaligned");
    const int words = (n + 63) / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, n), dim3(64), 0, (hipStream_t)stream, boxes, order, n, iou_threshold, (unsigned long long*)workspace, words);
    UMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)workspace, order, n, words, keep, n_keep);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_mask_paste_stats(const float* sdf_maps, const float* center_fields, const int32_t* boxes, int N, int S, int H, int W,
                                    int32_t* stats, float* maxima, umr_stream_t stream) {
    UMR_CHECK_ARG(sdf_maps && center_fields && boxes && stats && maxima && N > 0 && S > 0 && H > 0 && W > 0, "mask_paste_stats: bad arguments");
    if ((int64_t)S * S * 2 > 150 * 1024) return umr_set_error(UMR_ERR_UNSUPPORTED, "mask_paste_stats: crop larger than the LDS mask planes (S <= 277)");
    UMR_SET_MAX_LDS_ONCE(mask_paste_stats_kernel, 150 * 1024);
    hipLaunchKernelGGL(mask_paste_stats_kernel, dim3(N), dim3(256), (size_t)S * S * 2, (hipStream_t)stream, sdf_maps, center_fields, boxes, S, stats, maxima);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_mask_paste(const float* sdf_maps, const float* center_fields, const int32_t* boxes, const int64_t* select, int K, int S, int H, int W,
                              uint8_t* masks, umr_stream_t stream) {
    UMR_CHECK_ARG(sdf_maps && center_fields && boxes && select && masks && K > 0 && S > 0 && H > 0 && W > 0, "mask_paste: bad arguments");
    if ((int64_t)S * S * 2 > 150 * 1024) return umr_set_error(UMR_ERR_UNSUPPORTED, "mask_paste: crop larger than the LDS mask planes (S <= 277)");
    UMR_SET_MAX_LDS_ONCE(mask_paste_kernel, 150 * 1024);
    hipLaunchKernelGGL(mask_paste_kernel, dim3(K), dim3(256), (size_t)S * S * 2, (hipStream_t)stream, sdf_maps, center_fields, boxes, select, S, H, W, masks);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_mask_components(const float* sdf_maps, const float* center_fields, int B, int S, int max_components, int32_t* counts,
                                   int32_t* boxes, umr_stream_t stream) {
    UMR_CHECK_ARG(sdf_maps && center_fields && counts && boxes && B > 0 && S > 0 && max_components > 0, "mask_components: bad arguments");
    const size_t lds = (size_t)S * S * 8 + (size_t)max_components * 16;
    if (lds > 150 * 1024) return umr_set_error(UMR_ERR_UNSUPPORTED, "mask_components: S * S * 8 + max_components * 16 bytes of LDS exceed 150 KiB (S = 128: up to 1408 components)");
    UMR_SET_MAX_LDS_ONCE(mask_components_kernel, 150 * 1024);
    hipLaunchKernelGGL(mask_components_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, sdf_maps, center_fields, S, max_components, counts, boxes);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
