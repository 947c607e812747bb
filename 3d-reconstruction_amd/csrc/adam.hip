// adam.hip — one-launch Adam step over every parameter of the field.
// Replaces torch.optim.Adam(...).step() of train.py:272-273 / :376 (betas (0.9, 0.99), eps 1e-8, no weight
// decay, no amsgrad):   m <- b1 m + (1-b1) g,   v <- b2 v + (1-b2) g^2,
//                       p <- p - (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps).
// HBM-bound: 7 streams (p, m, v read+write, g read) of sum(numel) floats.  The parameter tensors, their
// gradients (views of the backward's contiguous gradient buffer) and the moment buffers are walked in STORAGE
// order (channel-last planes are dense in storage), 16 B per lane.  The segment table travels in the kernel
// arguments, so a hipGraph capture bakes the pointers in and nothing is read from host-updated tables; the
// learning rates and the step count are read from device memory (they change per step under graph replay).
#include "tf_device.h"

namespace {

constexpr int kChunk = TF_ADAM_CHUNK;   // elements per workgroup
static_assert(TF_ADAM_CHUNK == 8192, "the touched word holds 4 waves x 8 rounds of 256 floats");

// The launch's gate words, read ONCE and together (they are the same for every workgroup: scalar loads, one round trip —
// read one after the other behind branches they were two more L2 latencies in front of every workgroup's first load).
struct Gates {
    float live0, live1, over;      // density samples, shaded samples, overflow flag of the step (TfLive)
    float reg[4];                  // regulariser activity
};
__device__ __forceinline__ Gates load_gates(const TfAdamJob& J) {
    Gates g;
    g.live0 = J.live ? J.live[0] : 1.f;
    g.live1 = J.live ? J.live[1] : 1.f;
    g.over = J.live ? J.live[2] : 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) g.reg[b] = J.reg_active ? J.reg_active[b] : 0.f;
    return g;
}
// Is segment s updated by this launch?  (TfAdamJob: host skip mask, overflow, sample-count gate, regulariser flags.)
__device__ __forceinline__ bool seg_open(const TfAdamJob& J, const Gates& g, int s) {
    if ((J.skip_mask >> s) & 1u) return false;
    if (g.over != 0.f) return false;      // a right-sized list of this step ran full: its gradients are incomplete
    const int gate = J.seg[s].gate;
    const int cnt = gate & 3;
    if (!cnt) return true;
    if ((cnt == 1 ? g.live0 : g.live1) != 0.f) return true;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (((gate >> (4 + b)) & 1) && g.reg[b] != 0.f) return true;
    return false;
}

// Last workgroup of the launch: advance the counts of the updated segments, re-arm the arrival counter.
__device__ __forceinline__ void arrive(const TfAdamJob& J, const Gates& g) {
    if (atomicAdd(J.arrivals, 1u) == gridDim.x - 1) {
        for (int s = 0; s < J.n_seg; ++s)
            if (seg_open(J, g, s)) J.step[s] = J.step[s] + 1.f;
        *J.arrivals = 0u;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(const TfAdamJob J) {
    __shared__ float s_hyp[2];
    const int tid = threadIdx.x;
    // segment and gate of this workgroup: wave-uniform arithmetic on kernel arguments and a handful of scalar loads, done
    // by every thread (no LDS hand-over in front of the first memory request)
    const Gates gates = load_gates(J);
    // (counted over the whole table: 31 independent compares on two wide scalar loads — the `while` walk it replaces was a
    //  chain of up to n_seg dependent scalar loads in front of every workgroup's first memory request)
    int seg = 0;
#pragma unroll
    for (int s = 0; s < TF_ADAM_MAX_SEG - 1; ++s) seg += (s + 1 < J.n_seg && (int)blockIdx.x >= J.chunk_end[s]) ? 1 : 0;
    const bool open = seg_open(J, gates, seg);
    if (!open) {             // a parameter without a gradient this step: untouched, like torch.optim.Adam's `grad is None`
        // ... but a caller that accumulates the next step into the same buffer (clear_grads) must get it back clean: a
        // step whose lists overflowed leaves INCOMPLETE non-zero gradients behind, which are dropped here
        if (J.clear_grads && !((J.skip_mask >> seg) & 1u)) {
            const TfAdamSeg& sc = J.seg[seg];
            const long long c0 = (long long)((int)blockIdx.x - (seg ? J.chunk_end[seg - 1] : 0)) * kChunk;
            const long long n = sc.n - c0 < kChunk ? sc.n - c0 : kChunk;
            float* __restrict__ g = const_cast<float*>(sc.g) + c0;
            for (long long i = tid; i < n; i += 256)
                if (g[i] != 0.f) g[i] = 0.f;
        }
        __syncthreads();
        if (tid == 0) arrive(J, gates);
        return;
    }
    if (tid == 0) {
        const double t = (double)J.step[seg] + 1.0;    // step[s] counts the segment's completed updates; this is number t
        const double bc1 = 1.0 - pow(J.beta1, t), bc2 = 1.0 - pow(J.beta2, t);
        s_hyp[0] = (float)((double)J.lrs[J.seg[seg].group] / (double)(float)bc1);   // step size
        s_hyp[1] = (float)sqrt(bc2);
    }
    const int s_seg = seg;
    const TfAdamSeg& sg = J.seg[s_seg];
    float step_size = 0.f, bc2_sqrt = 1.f;      // read from LDS behind the first batch's loads (thread 0 is still computing them)
    const long long c0 = (long long)((int)blockIdx.x - (s_seg ? J.chunk_end[s_seg - 1] : 0)) * kChunk;
    const long long n = sg.n - c0 < kChunk ? sg.n - c0 : kChunk;
    float* __restrict__ p = sg.p + c0;
    float* __restrict__ g = const_cast<float*>(sg.g) + c0;      // (written only with J.clear_grads)
    float* __restrict__ m = sg.m + c0;
    float* __restrict__ v = sg.v + c0;
    const double b1 = J.beta1, b2 = J.beta2, eps = J.eps;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        mm = (float)(b1 * (double)mm + (1.0 - b1) * (double)gg);
        vv = (float)(b2 * (double)vv + (1.0 - b2) * (double)gg * (double)gg);
        const float denom = (float)((double)(sqrtf(vv) / bc2_sqrt) + eps);
        pp -= step_size * mm / denom;
    };
    const long long n4 = n >> 2;
    // `touched` (one word per workgroup): bit (8 * wave + round) is set once the 256 floats that wave handles in that
    // round have seen a non-zero gradient.  While it is clear their moments are still the zeros they were created
    // with, so only the gradient is read (1 of the 4 input streams); texels no sample ever reaches stay that way.
    // The 8 rounds of a thread go in two batches of 4 whose loads are all requested before the first is used (one round at
    // a time the kernel was a chain of 2 dependent memory latencies per round: 47 us at config 2 for ~130 MB of traffic).
    const int wave = tid >> 6;
    const unsigned seen = J.touched ? J.touched[blockIdx.x] : 0xFFFFFFFFu;
    unsigned fresh = 0;
    constexpr int kBatch = 4;
    static_assert(kChunk == 256 * 4 * 2 * kBatch, "two batches of kBatch rounds of 256 lanes x 4 floats");
    const tf::float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int b0 = 0; b0 < 2 * kBatch; b0 += kBatch) {
        tf::float4_t gv[kBatch], mv[kBatch], vv[kBatch], pv[kBatch];
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {      // every load whose need is known up front
            const long long i = tid + 256LL * (b0 + r);
            const bool in = i < n4, old = (seen >> (8 * wave + b0 + r)) & 1u;
            gv[r] = in ? tf::ld4(g + 4 * i) : zero4;
            mv[r] = in && old ? tf::ld4(m + 4 * i) : zero4;
            vv[r] = in && old ? tf::ld4(v + 4 * i) : zero4;
            pv[r] = in && old ? tf::ld4(p + 4 * i) : zero4;
        }
        if (b0 == 0) {
            __syncthreads();
            step_size = s_hyp[0];
            bc2_sqrt = s_hyp[1];
        }
        // Entries whose gradient and both moments are zero stay exactly as they are (m = v = 0, update 0 / (0 + eps)):
        // a wave that holds only such entries neither reads the parameters nor writes anything back, which is exact.
        bool live[kBatch];
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {
            bool l = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) l |= (gv[r][e] != 0.f) | (mv[r][e] != 0.f) | (vv[r][e] != 0.f);
            live[r] = __any(l);
            const long long i = tid + 256LL * (b0 + r);
            const bool old = (seen >> (8 * wave + b0 + r)) & 1u;
            if (live[r] && !old && i < n4) pv[r] = tf::ld4(p + 4 * i);     // first gradient of this piece: its parameters
        }
#pragma unroll
        for (int r = 0; r < kBatch; ++r) {
            const long long i = tid + 256LL * (b0 + r);
            if (!live[r] || i >= n4) continue;
            fresh |= (1u << (8 * wave + b0 + r)) & ~seen;
            tf::float4_t pq = pv[r], mq = mv[r], vq = vv[r];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pq[e], me = mq[e], ve = vq[e];
                upd(pe, gv[r][e], me, ve);
                pq[e] = pe; mq[e] = me; vq[e] = ve;
            }
            *reinterpret_cast<tf::float4_t*>(p + 4 * i) = pq;
            *reinterpret_cast<tf::float4_t*>(m + 4 * i) = mq;
            *reinterpret_cast<tf::float4_t*>(v + 4 * i) = vq;
            // consumed gradients go back to zero (lanes whose four are zero already write nothing; waves skipped above
            // hold zeros only)
            if (J.clear_grads && ((gv[r][0] != 0.f) | (gv[r][1] != 0.f) | (gv[r][2] != 0.f) | (gv[r][3] != 0.f)))
                *reinterpret_cast<tf::float4_t*>(g + 4 * i) = zero4;
        }
    }
    if (fresh && (tid & 63) == 0) atomicOr(J.touched + blockIdx.x, fresh);
    for (long long i = 4 * n4 + tid; i < n; i += 256) {
        upd(p[i], g[i], m[i], v[i]);
        if (J.clear_grads) g[i] = 0.f;
    }
    // every workgroup has read its step count by now or will have before it arrives here: the last one to arrive advances
    // the counts for the next launch and re-arms the arrival counter (no separate "step += 1" launch per update)
    __syncthreads();
    // (thread 0 consumed step[] before the first barrier of this workgroup: no fence needed, and a device-scope fence
    //  here would write back the L2 once per workgroup)
    if (tid == 0) arrive(J, gates);
}

}  // namespace

extern "C" int tf_adam_step(const TfAdamJob* job, tf_stream_t stream) {
    if (job->n_seg < 1 || job->n_seg > TF_ADAM_MAX_SEG || !job->lrs || !job->step || !job->arrivals)
        return (int)hipErrorInvalidValue;
    for (int s = 0; s < job->n_seg; ++s) {
        const TfAdamSeg& sg = job->seg[s];
        // 16-B lanes: every stream of a segment must be 16-B aligned
        if (((uintptr_t)sg.p | (uintptr_t)sg.g | (uintptr_t)sg.m | (uintptr_t)sg.v) & 15) return (int)hipErrorInvalidValue;
        const long long chunks = (sg.n + kChunk - 1) / kChunk;
        const int prev = s ? job->chunk_end[s - 1] : 0;
        if (sg.n <= 0 || job->chunk_end[s] - prev != chunks) return (int)hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)job->chunk_end[job->n_seg - 1]), dim3(256), 0, (hipStream_t)stream, *job);
    return TF_CHECK_LAUNCH();
}
