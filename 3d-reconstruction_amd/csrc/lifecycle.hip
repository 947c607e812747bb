// lifecycle.hip — the callers of the density lookup outside the per-ray march:
// compute_alpha on a point list (alpha-volume rebuild), AlphaGridMask.sample_alpha, filtering_rays.
// tensorBase.py:41-48, 215-230, 259-288, 298-318.   gfx950, wave64.
#include "tf_device.h"

using namespace tf;

namespace {

// alpha = 1 - exp(-sigma * length), sigma = feature2density(compute_densityfeature(normalize(p))) where the
// alpha mask (if any) is hit, else 0.   tensorBase.py:298-318.   4 lanes per point.
__global__ __launch_bounds__(256) void alpha_points_kernel(const TfField F, const float* __restrict__ xyz, int n,
                                                           float length, float* __restrict__ out) {
    const int groups = (n + 63) / 64;
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g < groups; g += gridDim.x * 4) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pt = g * 64 + s * 16 + (lane >> 2);
            float part = 0.f;
            bool hit = false;
            if (pt < n) {
                const float p[3] = {xyz[(size_t)pt * 3], xyz[(size_t)pt * 3 + 1], xyz[(size_t)pt * 3 + 2]};
                hit = F.alpha_cells == nullptr || alpha_hit(F, p);
                if (hit) {
                    float u[3];
                    normalize(F, p, u);
                    part = density_partial(F.model, F.density, F.grid, u, lane & 3);
                }
            }
            const float f = quad_sum(part);
            if (pt < n && (lane & 3) == 0) {
                const float sigma = hit ? density_act(F, f) : 0.f;
                out[pt] = 1.f - expf(-sigma * length);
            }
        }
    }
}

// F.grid_sample(volume (1,1,Gz,Gy,Gx), pts, align_corners=True, zeros padding), trilinear.  tensorBase.py:41-45
__global__ __launch_bounds__(256) void sample_alpha_kernel(const float* __restrict__ vol, int gx, int gy, int gz,
                                                           float lx, float ly, float lz, float ix, float iy, float iz,
                                                           const float* __restrict__ xyz, int n, float* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float lo[3] = {lx, ly, lz}, inv[3] = {ix, iy, iz};
        const int g[3] = {gx, gy, gz};
        int i0[3], i1[3];
        float w0[3], w1[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float t = xyz[(size_t)i * 3 + a] - lo[a];
            const float u = t * inv[a] - 1.f;
            tap_axis(u, g[a], i0[a], i1[a], w0[a], w1[a]);
        }
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int x = (c & 1) ? i1[0] : i0[0], y = (c & 2) ? i1[1] : i0[1], z = (c & 4) ? i1[2] : i0[2];
            const float w = ((c & 1) ? w1[0] : w0[0]) * ((c & 2) ? w1[1] : w0[1]) * ((c & 4) ? w1[2] : w0[2]);
            acc += vol[((size_t)z * gy + y) * gx + x] * w;
        }
        out[i] = acc;
    }
}

// filtering_rays (tensorBase.py:259-288): bbox_only -> t_max > t_min of the slab test (no near/far clamp);
// else -> any of the N eval samples of the ray hits the alpha mask.  One wave per ray.
__global__ __launch_bounds__(64) void filter_rays_kernel(const TfField F, const float* __restrict__ rays, int n_rays,
                                                         int bbox_only, int n_samples, uint8_t* __restrict__ keep) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_rays) return;
    Ray ray;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ray.o[a] = rays[(size_t)r * 6 + a];
        ray.d[a] = rays[(size_t)r * 6 + 3 + a];
    }
    ray.jit = 0.f;
    ray.dnorm = 1.f;
    if (bbox_only) {
        float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = (ray.d[a] == 0.f) ? 1e-6f : ray.d[a];
            const float ta = (F.aabb_hi[a] - ray.o[a]) / v, tb = (F.aabb_lo[a] - ray.o[a]) / v;
            tmin = fmaxf(tmin, fminf(ta, tb));
            tmax = fminf(tmax, fmaxf(ta, tb));
        }
        if (lane == 0) keep[r] = tmax > tmin;
        return;
    }
    ray.tmin = slab_tmin(F, ray);
    bool any = false;
    for (int base = 0; base < n_samples && !any; base += 64) {
        const int i = base + lane;
        bool hit = false;
        if (i < n_samples) {
            float p[3];
            sample_pos(ray, sample_z(F, ray, nullptr, i), p);
            hit = F.alpha_cells != nullptr && alpha_hit(F, p);
        }
        any = __ballot(hit) != 0;
    }
    if (lane == 0) keep[r] = any;
}

}  // namespace

extern "C" {

int tf_alpha_points(const TfField* field, const float* xyz, int n, float length, float* out_alpha, tf_stream_t stream) {
    if (n <= 0) return 0;
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(alpha_points_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *field, xyz, n, length,
                       out_alpha);
    return TF_CHECK_LAUNCH();
}

int tf_sample_alpha_points(const float* volume, int gx, int gy, int gz, const float lo[3], const float inv[3],
                           const float* xyz, int n, float* out, tf_stream_t stream) {
    if (n <= 0) return 0;
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sample_alpha_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, volume, gx, gy, gz, lo[0],
                       lo[1], lo[2], inv[0], inv[1], inv[2], xyz, n, out);
    return TF_CHECK_LAUNCH();
}

int tf_filter_rays(const TfField* field, const float* rays, int n_rays, int bbox_only, int n_samples, uint8_t* keep,
                   tf_stream_t stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(filter_rays_kernel, dim3(n_rays), dim3(64), 0, (hipStream_t)stream, *field, rays, n_rays,
                       bbox_only, n_samples, keep);
    return TF_CHECK_LAUNCH();
}

}  // extern "C"
