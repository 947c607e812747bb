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

// getDenseAlpha (tensorBase.py:215-230) without the (G^3, 3) point list: lattice node (ix, iy, iz) sits at
// aabb_lo (1 - s) + aabb_hi s with s = lin_a[i_a] — the caller's three 1-D torch.linspace(0, 1, G_a) tables, so the
// points are bit for bit the reference's — and its alpha goes to out[iz][iy][ix], the transposed layout
// updateAlphaMask continues with (:236-237).  4 lanes per node.
__global__ __launch_bounds__(256) void alpha_lattice_kernel(const TfField F, const float* __restrict__ lin_x,
                                                            const float* __restrict__ lin_y, const float* __restrict__ lin_z,
                                                            int gx, int gy, int gz, float length, float* __restrict__ out) {
    const long long n = (long long)gx * gy * gz;
    const long long groups = (n + 63) / 64;
    for (long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); g < groups; g += (long long)gridDim.x * 4) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long long node = g * 64 + s * 16 + (lane >> 2);          // = (iz * gy + iy) * gx + ix
            float part = 0.f;
            bool hit = false;
            if (node < n) {
                const int ix = (int)(node % gx), iy = (int)((node / gx) % gy), iz = (int)(node / ((long long)gx * gy));
                const float sv[3] = {lin_x[ix], lin_y[iy], lin_z[iz]};
                float p[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float lo = F.aabb_lo[a] * (1.f - sv[a]);          // aabb[0] * (1 - samples) + aabb[1] * samples
                    const float hi = F.aabb_hi[a] * sv[a];
                    p[a] = lo + hi;
                }
                hit = F.alpha_cells == nullptr || alpha_hit(F, p);
                if (hit) {
                    float u[3];
                    normalize(F, p, u);
                    part = density_partial(F.model, F.density, F.grid, u, lane & 3);
                }
            }
            const float f = quad_sum(part);
            if (node < n && (lane & 3) == 0) {
                const float sigma = hit ? density_act(F, f) : 0.f;
                out[node] = 1.f - expf(-sigma * length);
            }
        }
    }
}

// updateAlphaMask's tail (tensorBase.py:236-254) in one pass over the (gz, gy, gx) alpha volume: clamp to [0, 1], 3^3
// max-pool (stride 1, padding 1), threshold -> 0 / 1 volume; and, over the kept voxels, their count and the index box
// [min, max] per axis (the reference's amin / amax of the kept lattice points are the lattice coordinates at those
// indices: the tables are monotone).  stats: {count, min_x, min_y, min_z, max_x, max_y, max_z}, preset by the caller.
__global__ __launch_bounds__(256) void alpha_pool_kernel(const float* __restrict__ alpha, int gx, int gy, int gz, float thres,
                                                         float* __restrict__ vol, int* __restrict__ stats) {
    const long long n = (long long)gx * gy * gz;
    int cnt = 0, lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {-1, -1, -1};
    for (long long node = (long long)blockIdx.x * blockDim.x + threadIdx.x; node < n; node += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(node % gx), iy = (int)((node / gx) % gy), iz = (int)(node / ((long long)gx * gy));
        float m = -INFINITY;
        for (int dz = -1; dz <= 1; ++dz) {
            const int z = iz + dz;
            if (z < 0 || z >= gz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int y = iy + dy;
                if (y < 0 || y >= gy) continue;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int x = ix + dx;
                    if (x < 0 || x >= gx) continue;
                    const float a = fminf(fmaxf(alpha[((long long)z * gy + y) * gx + x], 0.f), 1.f);
                    m = fmaxf(m, a);
                }
            }
        }
        const bool keep = m >= thres;
        vol[node] = keep ? 1.f : 0.f;
        if (keep) {
            ++cnt;
            lo[0] = min(lo[0], ix); lo[1] = min(lo[1], iy); lo[2] = min(lo[2], iz);
            hi[0] = max(hi[0], ix); hi[1] = max(hi[1], iy); hi[2] = max(hi[2], iz);
        }
    }
    // wave reduction, then one atomic per wave and statistic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, 64);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = max(hi[a], __shfl_xor(hi[a], o, 64));
        }
    }
    if ((threadIdx.x & 63) == 0 && cnt > 0) {
        atomicAdd(&stats[0], cnt);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&stats[1 + a], lo[a]);
            atomicMax(&stats[4 + a], hi[a]);
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

// allrays[ray_idx], allrgbs[ray_idx] (train.py:297-298) in one launch: thread t moves float t of the 9 the batch row has
__global__ __launch_bounds__(256) void gather_batch_kernel(const float* __restrict__ rays, const float* __restrict__ rgbs,
                                                           const long long* __restrict__ ids, long long n_all, int n,
                                                           float* __restrict__ rays_out, float* __restrict__ rgbs_out,
                                                           const float* __restrict__ extra_src, float* __restrict__ extra_dst,
                                                           int n_extra, int n_gather_wgs, int pack_gx, const TfPackJob pack) {
    if ((int)blockIdx.x >= n_gather_wgs) {      // the workgroups behind the gather run the step's weight-pack job
        const int pb = (int)blockIdx.x - n_gather_wgs;
        tf::pack_block(pack, pb % pack_gx, pb / pack_gx, pack_gx);
        return;
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 9) {      // the tail of the grid moves the step's host-drawn numbers (pinned host memory, read in place)
        if (t - n * 9 < n_extra) extra_dst[t - n * 9] = extra_src[t - n * 9];
        return;
    }
    const int row = t / 9, c = t - row * 9;
    long long src = ids[row];
    if (src < 0) src += n_all;                       // torch indexing semantics for negative ids
    if ((unsigned long long)src >= (unsigned long long)n_all) return;   // out of range: row left untouched
    if (c < 6) rays_out[row * 6 + c] = rays[src * 6 + c];
    else rgbs_out[row * 3 + (c - 6)] = rgbs[src * 3 + (c - 6)];
}

// Camera rays for a list (or a contiguous range) of pixels.  dataLoader/ray_utils.py:24-63
// (pixel centre + 0.5, ((i - cx) / fx, +-(j - cy) / fy, +-1)), :66-87 (rays_d = dir @ c2w[:3,:3]^T, rays_o =
// c2w[:3,3]), dataLoader/blender.py:59 (directions normalised before the rotation), ray_utils.py:90-107 (NDC).
__global__ __launch_bounds__(256) void generate_rays_kernel(const TfCamera cam, const long long* __restrict__ ids,
                                                            long long first, int n, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const long long pix = ids ? ids[t] : first + t;
    const int j = (int)(pix / cam.width), i = (int)(pix - (long long)j * cam.width);
    float d[3];
    d[0] = ((float)i + 0.5f - cam.cx) / cam.fx;
    d[1] = ((float)j + 0.5f - cam.cy) / cam.fy;
    d[2] = 1.f;
    if (cam.opengl) {      // get_ray_directions_blender: y up, looking down -z
        d[1] = -d[1];
        d[2] = -1.f;
    }
    if (cam.normalize) {
        float q = d[0] * d[0];
        q = q + d[1] * d[1];
        q = q + d[2] * d[2];
        const float nrm = sqrtf(q);
        d[0] = d[0] / nrm; d[1] = d[1] / nrm; d[2] = d[2] / nrm;
    }
    float o[3], w[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* row = cam.c2w + 4 * r;
        float acc = d[0] * row[0];
        acc = acc + d[1] * row[1];
        acc = acc + d[2] * row[2];
        w[r] = acc;
        o[r] = row[3];
    }
    if (cam.ndc) {         // ndc_rays_blender
        const float tt = -(cam.ndc_near + o[2]) / w[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = o[r] + tt * w[r];
        const float sx = -1.f / (cam.width / (2.f * cam.fx)), sy = -1.f / (cam.height / (2.f * cam.fy));
        const float o0 = sx * o[0] / o[2], o1 = sy * o[1] / o[2], o2 = 1.f + 2.f * cam.ndc_near / o[2];
        const float d0 = sx * (w[0] / w[2] - o[0] / o[2]), d1 = sy * (w[1] / w[2] - o[1] / o[2]);
        const float d2 = -2.f * cam.ndc_near / o[2];
        o[0] = o0; o[1] = o1; o[2] = o2;
        w[0] = d0; w[1] = d1; w[2] = d2;
    }
    float* r6 = out + (size_t)t * 6;
    r6[0] = o[0]; r6[1] = o[1]; r6[2] = o[2];
    r6[3] = w[0]; r6[4] = w[1]; r6[5] = w[2];
}

}  // namespace

extern "C" {

int tf_gather_batch_staged(const float* rays, const float* rgbs, long long n_all, const long long* ids, int n, float* rays_out,
                           float* rgbs_out, const float* extra_src, float* extra_dst, int n_extra, const TfPackJob* pack,
                           tf_stream_t stream) {
    if (n < 0 || n_extra < 0 || (n_extra > 0 && (!extra_src || !extra_dst))) return (int)hipErrorInvalidValue;
    if (pack && (pack->n < 1 || pack->n > TF_PACK_MAX)) return (int)hipErrorInvalidValue;
    const int gather_wgs = (n * 9 + n_extra + 255) / 256;
    TfPackJob none{};
    const TfPackJob& pj = pack ? *pack : none;
    const int gx = pack ? tf::pack_grid_x(pj) : 1, gy = pack ? pj.n + ((pj.n_zero > 0 && pj.zero) ? 1 : 0) : 0;
    if (gather_wgs + gx * gy <= 0) return 0;
    hipLaunchKernelGGL(gather_batch_kernel, dim3(gather_wgs + gx * gy), dim3(256), 0, (hipStream_t)stream, rays, rgbs, ids,
                       n_all, n, rays_out, rgbs_out, extra_src, extra_dst, n_extra, gather_wgs, gx, pj);
    return TF_CHECK_LAUNCH();
}

int tf_gather_batch(const float* rays, const float* rgbs, long long n_all, const long long* ids, int n, float* rays_out,
                    float* rgbs_out, tf_stream_t stream) {
    if (n <= 0) return 0;
    return tf_gather_batch_staged(rays, rgbs, n_all, ids, n, rays_out, rgbs_out, nullptr, nullptr, 0, nullptr, stream);
}

int tf_generate_rays(const TfCamera* cam, const long long* pixel_ids, long long first_pixel, int n, float* rays_out,
                     tf_stream_t stream) {
    if (n <= 0) return 0;
    if (cam->width < 1 || cam->height < 1 || cam->fx == 0.f || cam->fy == 0.f) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(generate_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *cam, pixel_ids,
                       first_pixel, n, rays_out);
    return TF_CHECK_LAUNCH();
}


int tf_alpha_points(const TfField* field, const float* xyz, int n, float length, float* out_alpha, tf_stream_t stream) {
    if (n <= 0) return 0;
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(alpha_points_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *field, xyz, n, length,
                       out_alpha);
    return TF_CHECK_LAUNCH();
}

int tf_alpha_lattice(const TfField* field, const float* lin_x, const float* lin_y, const float* lin_z, int gx, int gy, int gz,
                     float length, float* out_alpha, tf_stream_t stream) {
    if (gx <= 0 || gy <= 0 || gz <= 0) return (int)hipErrorInvalidValue;
    const long long n = (long long)gx * gy * gz;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(alpha_lattice_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *field, lin_x, lin_y,
                       lin_z, gx, gy, gz, length, out_alpha);
    return TF_CHECK_LAUNCH();
}

int tf_alpha_pool_threshold(const float* alpha, int gx, int gy, int gz, float thres, float* volume, int* stats,
                            tf_stream_t stream) {
    if (gx <= 0 || gy <= 0 || gz <= 0 || !stats) return (int)hipErrorInvalidValue;
    const long long n = (long long)gx * gy * gz;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(alpha_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, alpha, gx, gy, gz, thres,
                       volume, stats);
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
