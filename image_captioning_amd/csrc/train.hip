// Kernels that exist only for the teacher-forced training step: the masked
// cross-entropy losses (forward and the gradient with respect to the logits) and
// the fused AdamW update over a flat parameter buffer.
#include "common.hpp"

namespace icap {

// Rows of a [G][gs][ld] tensor of which the first gr rows of every group are used
// (predictions[:, :-1] of a [B][T][V] tensor is gr = T-1, gs = T): row r lives at
// ((r / gr) * gs + r % gr) * ld.
__device__ __forceinline__ size_t grouped_row(int r, int gr, int gs, int ld) {
    const int g = r / gr;
    return ((size_t)g * gs + (r - g * gr)) * (size_t)ld;
}

// ---------------------------------------------------------------- cross entropy, forward
// One workgroup per row: lse = log sum exp(logits[r]), loss = lse - logits[r][target] (0 for ignored rows).
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, int ld, int gr, int gs,
                                                      const long long* __restrict__ targets, int V, long long ignore,
                                                      float* __restrict__ row_loss, float* __restrict__ row_lse) {
    __shared__ float red[4];
    const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* x = logits + grouped_row(r, gr, gs, ld);
    float m = -INFINITY;
    for (int v = tid; v < V; v += 256) m = fmaxf(m, x[v]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int v = tid; v < V; v += 256) s += expf(x[v] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        const float lse = m + logf(((red[0] + red[1]) + red[2]) + red[3]);
        const long long t = targets[r];
        row_lse[r] = lse;
        row_loss[r] = (t == ignore || t < 0 || t >= V) ? 0.f : lse - x[t];
    }
}

// Deterministic reduction of the row losses: out[0] = loss, out[1] = the scale every row was weighted with
// (1/R when pads stay in the denominator, 1/count of non-ignored rows otherwise; 0 rows -> scale 0).
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ row_loss,
                                                        const long long* __restrict__ targets, int R, int V,
                                                        long long ignore, int mean_over_valid, float* __restrict__ out) {
    __shared__ float rs[4];
    __shared__ float rc[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float s = 0.f, c = 0.f;
    for (int r = tid; r < R; r += 256) {
        s += row_loss[r];
        const long long t = targets[r];
        c += (t == ignore || t < 0 || t >= V) ? 0.f : 1.f;
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if (lane == 0) { rs[wave] = s; rc[wave] = c; }
    __syncthreads();
    if (tid == 0) {
        const float sum = ((rs[0] + rs[1]) + rs[2]) + rs[3];
        const float cnt = ((rc[0] + rc[1]) + rc[2]) + rc[3];
        const float scale = mean_over_valid ? (cnt > 0.f ? 1.0f / cnt : 0.f) : 1.0f / (float)R;
        out[0] = sum * scale;
        out[1] = scale;
    }
}

int ce_loss_fwd_f32(const float* logits, int ld, int group_rows, int group_stride, const long long* targets, int R, int V,
                    long long ignore_index, int mean_over_valid, float* row_loss, float* row_lse, float* out,
                    hipStream_t st) {
    if (!logits || !targets || !row_loss || !row_lse || !out || R <= 0 || V <= 0 || ld < V || group_rows <= 0 ||
        group_stride < group_rows) {
        set_error("ce_loss_fwd: bad arguments");
        return ICAP_ERR_ARG;
    }
    hipLaunchKernelGGL(ce_rows_kernel, dim3(R), dim3(256), 0, st, logits, ld, group_rows, group_stride, targets, V, ignore_index, row_loss, row_lse);
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, st, row_loss, targets, R, V, ignore_index, mean_over_valid, out);
    return launch_status("ce_loss_fwd");
}

// ---------------------------------------------------------------- cross entropy, gradient
// dlogits[r][v] = (softmax(logits[r])[v] - [v == target]) * grad_out * scale, 0 for ignored rows;
// dlogits is a dense [R][ldd] matrix whatever the row grouping of the logits.
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int ld, int gr, int gs,
                                                     const long long* __restrict__ targets, int V, long long ignore,
                                                     const float* __restrict__ row_lse, const float* __restrict__ meta,
                                                     const float* __restrict__ grad_out, float* __restrict__ dlogits,
                                                     int ldd) {
    const int r = blockIdx.x;
    const long long t = targets[r];
    const bool ignored = (t == ignore || t < 0 || t >= V);
    const float* x = logits + grouped_row(r, gr, gs, ld);
    float* d = dlogits + (size_t)r * ldd;
    const float g = grad_out[0] * meta[1];
    const float lse = row_lse[r];
    for (int v = blockIdx.y * 256 + threadIdx.x; v < V; v += 256 * gridDim.y) {
        float val = 0.f;
        if (!ignored) val = (expf(x[v] - lse) - (v == (int)t ? 1.f : 0.f)) * g;
        d[v] = val;
    }
}

int ce_loss_bwd_f32(const float* logits, int ld, int group_rows, int group_stride, const long long* targets, int R, int V,
                    long long ignore_index, const float* row_lse, const float* meta, const float* grad_out, float* dlogits,
                    int ldd, hipStream_t st) {
    if (!logits || !targets || !row_lse || !meta || !grad_out || !dlogits || R <= 0 || V <= 0 || ld < V || ldd < V ||
        group_rows <= 0 || group_stride < group_rows) {
        set_error("ce_loss_bwd: bad arguments");
        return ICAP_ERR_ARG;
    }
    const int ychunks = V > 2048 ? 4 : 1;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(R, ychunks), dim3(256), 0, st, logits, ld, group_rows, group_stride, targets, V, ignore_index, row_lse, meta, grad_out, dlogits, ldd);
    return launch_status("ce_loss_bwd");
}

// ---------------------------------------------------------------- AdamW
// Decoupled weight decay, the update order of torch.optim.AdamW:
//   p *= 1 - lr*wd;  m = m + (g - m)*(1 - b1);  v = v*b2 + g*g*(1 - b2);
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// over one flat buffer (all parameters of the model live in it); n % 4 == 0.
__global__ __launch_bounds__(256) void adamw_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                    float4* __restrict__ m, float4* __restrict__ v, long long n4,
                                                    float decay, float one_minus_b1, float b2, float one_minus_b2,
                                                    float step_size, float inv_bc2_sqrt, float eps, float grad_scale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float* pa = reinterpret_cast<float*>(&pp);
        float* ga = reinterpret_cast<float*>(&gg);
        float* ma = reinterpret_cast<float*>(&mm);
        float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = ga[k] * grad_scale;
            float pk = pa[k] * decay;
            const float mk = ma[k] + (gr - ma[k]) * one_minus_b1;
            const float vk = va[k] * b2 + gr * gr * one_minus_b2;
            const float denom = sqrtf(vk) * inv_bc2_sqrt + eps;
            pk -= step_size * (mk / denom);
            pa[k] = pk; ma[k] = mk; va[k] = vk;
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

// The same update with the step count t = *step_dev + step_bias read from device memory (a launch recorded in a HIP
// graph keeps its arguments; the counter is what advances between replays).  The bias corrections are computed per
// lane in double precision, as the host does for the launch-argument form.
__global__ __launch_bounds__(256) void adamw_dev_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                        float4* __restrict__ m, float4* __restrict__ v, long long n4,
                                                        float lr, float decay, float b1, float b2, float eps, float grad_scale,
                                                        const int* __restrict__ step_dev, int step_bias) {
    const double t = (double)(*step_dev + step_bias);
    const float step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    const float one_minus_b1 = 1.0f - b1, one_minus_b2 = 1.0f - b2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float* pa = reinterpret_cast<float*>(&pp);
        float* ga = reinterpret_cast<float*>(&gg);
        float* ma = reinterpret_cast<float*>(&mm);
        float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = ga[k] * grad_scale;
            float pk = pa[k] * decay;
            const float mk = ma[k] + (gr - ma[k]) * one_minus_b1;
            const float vk = va[k] * b2 + gr * gr * one_minus_b2;
            const float denom = sqrtf(vk) * inv_bc2_sqrt + eps;
            pk -= step_size * (mk / denom);
            pa[k] = pk; ma[k] = mk; va[k] = vk;
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

int adamw_dev_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, const int* step_dev, int step_bias, float grad_scale, hipStream_t st) {
    if (!p || !g || !m || !v || !step_dev || n <= 0) { set_error("adamw_dev: bad arguments"); return ICAP_ERR_ARG; }
    if (n % 4 || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) {
        set_error("adamw_dev: n %% 4 and 16-byte aligned buffers required");
        return ICAP_ERR_ALIGN;
    }
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(p),
                       reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), n4,
                       lr, 1.0f - lr * weight_decay, beta1, beta2, eps, grad_scale, step_dev, step_bias);
    return launch_status("adamw_dev");
}

int adamw_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, float grad_scale, hipStream_t st) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) { set_error("adamw: bad arguments"); return ICAP_ERR_ARG; }
    if (n % 4 || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) {
        set_error("adamw: n %% 4 and 16-byte aligned buffers required");
        return ICAP_ERR_ALIGN;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(p),
                       reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), n4,
                       1.0f - lr * weight_decay, 1.0f - beta1, beta2, 1.0f - beta2, step_size, inv_bc2_sqrt, eps, grad_scale);
    return launch_status("adamw");
}

}  // namespace icap
