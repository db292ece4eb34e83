// C-ABI entry points of libsgpr_hip.so (declared in include/sgpr.h) and the host-side
// weight preparation: eval-mode BatchNorm folding + kernel-ready layout.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <exception>
#include <string>
#include <vector>

#include "sgpr_internal.hpp"

namespace sgpr {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int raise_lds_limit(LdsLimitOnce* once, const void* kernel, int bytes, const char* what) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (once->done.load(std::memory_order_acquire) & bit)) return SGPR_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        int have = 0;
        (void)hipDeviceGetAttribute(&have, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        set_error(std::string(what) + ": the kernel needs " + std::to_string(bytes) + " bytes of LDS per workgroup, device " +
                  std::to_string(dev) + " offers " + std::to_string(have) + " (" + hipGetErrorString(e) + ")");
        return SGPR_E_HIP;
    }
    if (dev < 64) once->done.fetch_or(bit, std::memory_order_release);
    return SGPR_OK;
}

int hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return SGPR_E_HIP;
}

// The kernels are written for {12, 64, 64, 32, 16, 16}.  A SMALLER architecture is served exactly by the same kernels:
// its tensors are embedded into the built shapes with zero weights (and neutral BatchNorm statistics) for the channels
// it does not have - a channel whose weights are all zero stays 0 through conv / BN / LeakyReLU, adds 0 to every
// distance, every sum and every maximum partner, and a zero tensor / bottleneck neuron contributes 0 to the score.
static bool dims_supported(const sgpr_dims* d) {
    return d && d->num_labels >= 1 && d->num_labels <= kLabels && d->filters_1 >= 1 && d->filters_1 <= kF1 &&
           d->filters_2 >= 1 && d->filters_2 <= kF2 && d->filters_3 >= 1 && d->filters_3 <= kF3 &&
           d->tensor_neurons >= 1 && d->tensor_neurons <= kT && d->bottle_neck_neurons >= 1 && d->bottle_neck_neurons <= kB;
}
static bool dims_full(const sgpr_dims* d) {
    return d->num_labels == kLabels && d->filters_1 == kF1 && d->filters_2 == kF2 && d->filters_3 == kF3 &&
           d->tensor_neurons == kT && d->bottle_neck_neurons == kB;
}

// what the any-shape kernels (sgpr_generic.hip) serve
static bool dims_generic(const sgpr_dims* d) {
    return d && d->num_labels >= 1 && d->num_labels <= SGPR_GENERIC_MAX_LABELS && d->filters_1 >= 1 &&
           d->filters_1 <= SGPR_GENERIC_MAX_FILTERS && d->filters_2 >= 1 && d->filters_2 <= SGPR_GENERIC_MAX_FILTERS &&
           d->filters_3 >= 1 && d->filters_3 <= SGPR_GENERIC_MAX_F3 && d->tensor_neurons >= 1 &&
           d->tensor_neurons <= SGPR_GENERIC_MAX_T && d->bottle_neck_neurons >= 1 && d->bottle_neck_neurons <= SGPR_GENERIC_MAX_T;
}

struct BlockShape {
    int cout, cin2;
};

static void block_shapes(const sgpr_dims* d, BlockShape out[7]) {
    // order of the blob: s_conv1, f_conv1, s_conv2, f_conv2, s_conv3, f_conv3, conv_end (sg_net.py:50-76)
    out[0] = {d->filters_1, 6};
    out[1] = {d->filters_1, 2 * d->num_labels};
    out[2] = {d->filters_2, 2 * d->filters_1};
    out[3] = {d->filters_2, 2 * d->filters_1};
    out[4] = {d->filters_3, 2 * d->filters_2};
    out[5] = {d->filters_3, 2 * d->filters_2};
    out[6] = {d->filters_3, 2 * d->filters_3};
}

static size_t weights_count(const sgpr_dims* d);
// the blob of a smaller architecture `d` scattered into the built architecture's blob layout
static std::vector<float> pad_blob(const float* src, const sgpr_dims* d) {
    const sgpr_dims full = {kLabels, kF1, kF2, kF3, kT, kB};
    BlockShape bs[7], bf[7];
    block_shapes(d, bs);
    block_shapes(&full, bf);
    std::vector<float> out(weights_count(&full), 0.f);
    float* dst = out.data();
    for (int b = 0; b < 7; ++b) {
        const int co = bs[b].cout, ci = bs[b].cin2 / 2, CO = bf[b].cout, CI = bf[b].cin2 / 2;
        // weight [cout][2 cin]: the (x_j - x_i) half and the x_i half keep their places in the wider halves (conv_end:
        // the xyz3 half and the sem3 half of cat(xyz3, sem3))
        for (int c = 0; c < co; ++c)
            for (int i = 0; i < ci; ++i) {
                dst[(size_t)c * 2 * CI + i] = src[(size_t)c * 2 * ci + i];
                dst[(size_t)c * 2 * CI + CI + i] = src[(size_t)c * 2 * ci + ci + i];
            }
        src += (size_t)co * 2 * ci;
        dst += (size_t)CO * 2 * CI;
        const float neutral[4] = {1.f, 0.f, 0.f, 1.f};          // gamma, beta, running_mean, running_var
        for (int v = 0; v < 4; ++v) {
            for (int c = 0; c < CO; ++c) dst[c] = c < co ? src[c] : neutral[v];
            src += co;
            dst += CO;
        }
    }
    const int f = d->filters_3, t = d->tensor_neurons, bn = d->bottle_neck_neurons;
    for (int i = 0; i < f; ++i)                                  // attention.weight_matrix [F3][F3]
        for (int j = 0; j < f; ++j) dst[(size_t)i * kF3 + j] = src[(size_t)i * f + j];
    src += (size_t)f * f;
    dst += (size_t)kF3 * kF3;
    for (int i = 0; i < f; ++i)                                  // tensor_network.weight_matrix [F3][F3][T]
        for (int j = 0; j < f; ++j)
            for (int q = 0; q < t; ++q) dst[((size_t)i * kF3 + j) * kT + q] = src[((size_t)i * f + j) * t + q];
    src += (size_t)f * f * t;
    dst += (size_t)kF3 * kF3 * kT;
    for (int q = 0; q < t; ++q)                                  // weight_matrix_block [T][2 F3]
        for (int j = 0; j < f; ++j) {
            dst[(size_t)q * 2 * kF3 + j] = src[(size_t)q * 2 * f + j];
            dst[(size_t)q * 2 * kF3 + kF3 + j] = src[(size_t)q * 2 * f + f + j];
        }
    src += (size_t)t * 2 * f;
    dst += (size_t)kT * 2 * kF3;
    for (int q = 0; q < t; ++q) dst[q] = src[q];                 // tensor_network.bias [T]
    src += t;
    dst += kT;
    for (int o = 0; o < bn; ++o)                                 // fully_connected_first.weight [B][T]
        for (int q = 0; q < t; ++q) dst[(size_t)o * kT + q] = src[(size_t)o * t + q];
    src += (size_t)bn * t;
    dst += (size_t)kB * kT;
    for (int o = 0; o < bn; ++o) dst[o] = src[o];                // fully_connected_first.bias [B]
    src += bn;
    dst += kB;
    for (int o = 0; o < bn; ++o) dst[o] = src[o];                // scoring_layer.weight [1][B]
    src += bn;
    dst += kB;
    dst[0] = src[0];                                             // scoring_layer.bias
    return out;
}

static size_t weights_count(const sgpr_dims* d) {
    BlockShape bs[7];
    block_shapes(d, bs);
    size_t n = 0;
    for (int b = 0; b < 7; ++b) n += (size_t)bs[b].cout * bs[b].cin2 + 4 * (size_t)bs[b].cout;
    const size_t f = d->filters_3, t = d->tensor_neurons, bn = d->bottle_neck_neurons;
    n += f * f;                  // attention.weight_matrix
    n += f * f * t + t * 2 * f + t;  // tensor_network
    n += bn * t + bn;            // fully_connected_first
    n += bn + 1;                 // scoring_layer
    return n;
}

}  // namespace sgpr

using namespace sgpr;

extern "C" {

size_t sgpr_weights_count(const sgpr_dims* dims) { return dims ? weights_count(dims) : 0; }

int sgpr_abi_version(void) { return SGPR_ABI_VERSION; }

const char* sgpr_last_error(void) { return g_last_error.c_str(); }

// w = hi + mid + lo with three round-to-nearest-even bf16 terms (exact to 24 bits)
static unsigned short bf16_rne_host(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_to_float_host(unsigned short h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void split3_host(float w, unsigned short (&pl)[3]) {
    pl[0] = bf16_rne_host(w);
    float r = w - bf16_to_float_host(pl[0]);
    pl[1] = bf16_rne_host(r);
    r -= bf16_to_float_host(pl[1]);
    pl[2] = bf16_rne_host(r);
}

#ifndef SGPR_WIDE_EMBED
#define SGPR_WIDE_EMBED 1     // any-shape handles inside sgpr_wide.hip's limits embed on the matrix cores (0: A/B builds, plain fp32 only)
#endif
// The any-shape model of a blob (GenericModel): BatchNorm folded in double precision at the model's own dimensions,
// uploaded as one allocation.  Blob order: s_conv1, f_conv1, s_conv2, f_conv2, s_conv3, f_conv3, conv_end, tail tensors.
static int build_generic_model(const float* weights, const sgpr_dims* d, sgpr_handle* h) {
    BlockShape bs[7];
    block_shapes(d, bs);
    std::vector<float> host;
    size_t off_wa[6], off_wb[6], off_tb[6], off_wend = 0, off_tend = 0;
    const float* src = weights;
    GenericModel& m = h->gm;
    memset(&m, 0, sizeof(m));
    m.L = d->num_labels; m.f1 = d->filters_1; m.f2 = d->filters_2; m.f3 = d->filters_3;
    m.T = d->tensor_neurons; m.B = d->bottle_neck_neurons;
    m.cmax = std::max(std::max(3, m.L), std::max(m.f1, std::max(m.f2, m.f3)));
    for (int b = 0; b < 7; ++b) {
        const int cout = bs[b].cout, cin2 = bs[b].cin2, cin = cin2 / 2;
        const float* W = src;
        const float* gamma = W + (size_t)cout * cin2;
        const float* beta = gamma + cout;
        const float* mean = beta + cout;
        const float* var = mean + cout;
        src = var + cout;
        if (b < 6) {
            // blob order alternates the branches (s1, f1, s2, f2, s3, f3); the model keeps xyz layers 0..2, semantic 3..5
            const int l6 = (b & 1) * 3 + (b >> 1);
            m.cin[l6] = cin;
            m.cout[l6] = cout;
            const int cout8 = (cout + 7) & ~7;                      // [cin][cout8]: see GenericModel
            off_wa[l6] = host.size();
            host.resize(host.size() + (size_t)cout8 * cin, 0.f);
            off_wb[l6] = host.size();
            host.resize(host.size() + (size_t)cout8 * cin, 0.f);
            off_tb[l6] = host.size();
            host.resize(host.size() + cout8, 0.f);
            for (int c = 0; c < cout; ++c) {
                const double s = (double)gamma[c] / std::sqrt((double)var[c] + 1e-5);
                for (int i = 0; i < cin; ++i) {
                    const double w1 = W[(size_t)c * cin2 + i], w2 = W[(size_t)c * cin2 + cin + i];
                    host[off_wa[l6] + (size_t)i * cout8 + c] = (float)(s * w1);            // acts on x_j
                    host[off_wb[l6] + (size_t)i * cout8 + c] = (float)(s * (w2 - w1));     // acts on x_i
                }
                host[off_tb[l6] + c] = (float)((double)beta[c] - (double)mean[c] * s);
            }
        } else {
            const int cout8 = (cout + 7) & ~7;
            off_wend = host.size();
            host.resize(host.size() + (size_t)cout8 * cin2, 0.f);
            off_tend = host.size();
            host.resize(host.size() + cout8, 0.f);
            for (int c = 0; c < cout; ++c) {
                const double s = (double)gamma[c] / std::sqrt((double)var[c] + 1e-5);
                for (int i = 0; i < cin2; ++i) host[off_wend + (size_t)i * cout8 + c] = (float)(s * W[(size_t)c * cin2 + i]);
                host[off_tend + c] = (float)((double)beta[c] - (double)mean[c] * s);
            }
        }
    }
    const size_t f = m.f3, t = m.T, bn = m.B;
    const size_t off_tail = host.size();
    const size_t n_tail = f * f + f * f * t + t * 2 * f + t + bn * t + bn + bn + 1;
    host.insert(host.end(), src, src + n_tail);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->d_gblob), host.size() * sizeof(float));
    if (e != hipSuccess) return hip_fail(e, "sgpr_create: any-shape model");
    e = hipMemcpy(h->d_gblob, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "sgpr_create: any-shape model upload");
    const float* dv = h->d_gblob;
    for (int l = 0; l < 6; ++l) {
        m.wa[l] = dv + off_wa[l];
        m.wb[l] = dv + off_wb[l];
        m.tb[l] = dv + off_tb[l];
    }
    m.w_end = dv + off_wend;
    m.t_end = dv + off_tend;
    const float* q = dv + off_tail;
    m.att_w = q;    q += f * f;
    m.ntn_w = q;    q += f * f * t;
    m.ntn_wb = q;   q += t * 2 * f;
    m.ntn_bias = q; q += t;
    m.fc1_w = q;    q += bn * t;
    m.fc1_b = q;    q += bn;
    m.fc2_w = q;    q += bn;
    m.fc2_b = q;
    // ---- the matrix-core form of the same folded weights (WideModel, sgpr_wide.hip) for moderately larger architectures:
    //      widths padded to 32 with zeros, two f16 planes (w = hi + lo, 22 bits) in MFMA operand order.  Only for an
    //      any-shape handle (the built shape runs on the tuned kernels) inside the limits and the f16 range.
    WideModel& wm = h->wm;
    memset(&wm, 0, sizeof(wm));
    h->d_wblob = nullptr;
    float wmax = 0.f;
    for (size_t i = 0; i < off_tail; ++i) wmax = std::max(wmax, fabsf(host[i]));
    if (h->generic_only && m.L <= SGPR_WIDE_MAX_LABELS && m.f1 <= SGPR_WIDE_MAX_FILTERS && m.f2 <= SGPR_WIDE_MAX_FILTERS &&
        m.f3 <= SGPR_WIDE_MAX_F3 && wmax < 60000.f && SGPR_WIDE_EMBED) {
        auto p32 = [](int v) { return (v + 31) & ~31; };
        std::vector<unsigned short> planes;
        std::vector<float> tbs;
        size_t off_wh[7], off_tbp[7];
        auto put = [&planes](float v, size_t at_hi, size_t at_lo) {
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            memcpy(&planes[at_hi], &hi, 2);
            memcpy(&planes[at_lo], &lo, 2);
        };
        for (int l = 0; l < 6; ++l) {
            const int cin = m.cin[l], cout = m.cout[l], cout8 = (cout + 7) & ~7;
            const int cinP = p32(cin), coutP = p32(cout), nct = 2 * coutP / 16, ks = cinP / 32;
            wm.cinP[l] = cinP;
            wm.coutP[l] = coutP;
            off_wh[l] = planes.size();
            planes.resize(planes.size() + (size_t)nct * ks * 2 * 512, 0);
            off_tbp[l] = tbs.size();
            tbs.resize(tbs.size() + coutP, 0.f);
            for (int c = 0; c < cout; ++c) tbs[off_tbp[l] + c] = host[off_tb[l] + c];
            for (int ct = 0; ct < nct; ++ct)
                for (int st = 0; st < ks; ++st)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e8 = 0; e8 < 8; ++e8) {
                            const int row = ct * 16 + (lane & 15), ch = 32 * st + 8 * (lane >> 4) + e8;
                            const bool brow = row >= coutP;                         // b rows follow the a rows
                            const int oc = brow ? row - coutP : row;
                            float v = 0.f;
                            if (oc < cout && ch < cin) v = host[(brow ? off_wb[l] : off_wa[l]) + (size_t)ch * cout8 + oc];
                            const size_t base = off_wh[l] + ((size_t)(ct * ks + st) * 2) * 512 + (size_t)lane * 8 + e8;
                            put(v, base, base + 512);
                        }
        }
        {   // conv_end on cat(xyz3 [F3P], sem3 [F3P]): input channel F3P + c is the reference's f3 + c
            const int f3 = m.f3, F3P = p32(f3), f8 = (f3 + 7) & ~7, nct = F3P / 16, ks = 2 * F3P / 32;
            wm.F3P = F3P;
            off_wh[6] = planes.size();
            planes.resize(planes.size() + (size_t)nct * ks * 2 * 512, 0);
            off_tbp[6] = tbs.size();
            tbs.resize(tbs.size() + F3P, 0.f);
            for (int c = 0; c < f3; ++c) tbs[off_tbp[6] + c] = host[off_tend + c];
            for (int ct = 0; ct < nct; ++ct)
                for (int st = 0; st < ks; ++st)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e8 = 0; e8 < 8; ++e8) {
                            const int oc = ct * 16 + (lane & 15), ch = 32 * st + 8 * (lane >> 4) + e8;
                            const int half = ch >= F3P ? 1 : 0, c = ch - half * F3P;
                            float v = 0.f;
                            if (oc < f3 && c < f3) v = host[off_wend + (size_t)(half * f3 + c) * f8 + oc];
                            const size_t base = off_wh[6] + ((size_t)(ct * ks + st) * 2) * 512 + (size_t)lane * 8 + e8;
                            put(v, base, base + 512);
                        }
        }
        const size_t plane_bytes = (planes.size() * 2 + 255) & ~(size_t)255;
        e = hipMalloc(&h->d_wblob, plane_bytes + tbs.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->d_wblob, planes.data(), planes.size() * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = hipMemcpy(static_cast<char*>(h->d_wblob) + plane_bytes, tbs.data(), tbs.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (h->d_wblob) (void)hipFree(h->d_wblob);
            h->d_wblob = nullptr;
            return hip_fail(e, "sgpr_create: matrix-core form of the any-shape model");
        }
        const unsigned short* pv = static_cast<const unsigned short*>(h->d_wblob);
        const float* tv = reinterpret_cast<const float*>(static_cast<const char*>(h->d_wblob) + plane_bytes);
        for (int l = 0; l < 6; ++l) {
            wm.wh[l] = pv + off_wh[l];
            wm.tbp[l] = tv + off_tbp[l];
        }
        wm.wh_end = pv + off_wh[6];
        wm.tbp_end = tv + off_tbp[6];
        wm.att_w = m.att_w;
        wm.L = m.L;
        wm.f3 = m.f3;
        wm.ok = 1;
    }
    return SGPR_OK;
}

int sgpr_create(const float* weights, size_t n_floats, const sgpr_dims* dims, int device, sgpr_handle** out) {
    if (!weights || !dims || !out) {
        set_error("sgpr_create: NULL argument");
        return SGPR_E_INVALID;
    }
    if (!dims_supported(dims) && !dims_generic(dims)) {
        set_error("sgpr_create: architecture outside what the any-shape kernels serve (labels <= 64, filters <= 256, "
                  "filters_3 <= 128, tensor / bottleneck neurons <= 64)");
        return SGPR_E_DIMS;
    }
    if (n_floats != weights_count(dims)) {
        set_error("sgpr_create: weights blob has " + std::to_string(n_floats) + " floats, expected " +
                  std::to_string(weights_count(dims)));
        return SGPR_E_BLOB;
    }
    if (!dims_supported(dims)) {
        // larger than the shape the tuned kernels are built for {labels 12, filters 64/64/32, tensor 16, bottleneck 16}:
        // the handle runs every call on the any-shape kernels (sgpr_generic.hip)
        DeviceGuard guard(device);
        sgpr_handle* h = new sgpr_handle();
        memset(static_cast<void*>(h), 0, sizeof(*h));
        h->device = device;
        h->dims = *dims;
        h->generic_only = 1;
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_status), sizeof(int32_t));
        if (e == hipSuccess) e = hipMemset(h->d_status, 0, sizeof(int32_t));
        int rc = e == hipSuccess ? build_generic_model(weights, dims, h) : hip_fail(e, "sgpr_create");
        if (rc != SGPR_OK) {
            if (h->d_status) (void)hipFree(h->d_status);
            if (h->d_gblob) (void)hipFree(h->d_gblob);
            delete h;
            return rc;
        }
        h->num_cus = prop.multiProcessorCount;
        *out = h;
        return SGPR_OK;
    }
    const float* user_weights = weights;
    const sgpr_dims user_dims = *dims;
    const sgpr_dims full_dims = {kLabels, kF1, kF2, kF3, kT, kB};
    std::vector<float> padded;
    if (!dims_full(dims)) {                  // a smaller architecture inside the built one (dims_supported)
        padded = pad_blob(weights, dims);
        weights = padded.data();
        n_floats = padded.size();
        dims = &full_dims;
    }
    BlockShape bs[7];
    block_shapes(dims, bs);

    // ---- fold BN (double precision) and lay out: 6 EdgeConv layers + conv_end + tail tensors
    std::vector<float> packed;
    size_t off_wf[7], off_tb[7];
    int kp_of[7];
    const float* src = weights;
    for (int b = 0; b < 7; ++b) {
        const int cout = bs[b].cout, cin2 = bs[b].cin2;
        const float* W = src;
        const float* gamma = W + (size_t)cout * cin2;
        const float* beta = gamma + cout;
        const float* mean = beta + cout;
        const float* var = mean + cout;
        src = var + cout;
        if (b < 6) {
            const int cin = cin2 / 2;
            const int kp = cin <= kKPad ? kKPad : cin;
            kp_of[b] = kp;
            off_wf[b] = packed.size();
            packed.resize(packed.size() + (size_t)2 * cout * kp, 0.f);
            float* wf = packed.data() + off_wf[b];
            for (int c = 0; c < cout; ++c) {
                const double s = (double)gamma[c] / sqrt((double)var[c] + 1e-5);
                for (int i = 0; i < cin; ++i) {
                    const double w1 = W[(size_t)c * cin2 + i];        // multiplies (x_j - x_i)   dgcnn.py:47
                    const double w2 = W[(size_t)c * cin2 + cin + i];  // multiplies x_i
                    wf[(size_t)c * kp + i] = (float)(s * w1);
                    wf[(size_t)(cout + c) * kp + i] = (float)(s * (w2 - w1));
                }
            }
        } else {
            kp_of[b] = cin2;
            off_wf[b] = packed.size();
            packed.resize(packed.size() + (size_t)cout * cin2, 0.f);
            float* wf = packed.data() + off_wf[b];
            for (int c = 0; c < cout; ++c) {
                const double s = (double)gamma[c] / sqrt((double)var[c] + 1e-5);
                for (int i = 0; i < cin2; ++i) wf[(size_t)c * cin2 + i] = (float)(s * W[(size_t)c * cin2 + i]);
            }
        }
        off_tb[b] = packed.size();
        packed.resize(packed.size() + cout, 0.f);
        float* tb = packed.data() + off_tb[b];
        for (int c = 0; c < cout; ++c) {
            const double s = (double)gamma[c] / sqrt((double)var[c] + 1e-5);
            tb[c] = (float)((double)beta[c] - (double)mean[c] * s);
        }
    }
    const size_t f = kF3, t = kT, bn = kB;
    auto append = [&](size_t n) {
        size_t o = packed.size();
        packed.insert(packed.end(), src, src + n);
        src += n;
        return o;
    };
    const size_t o_att = append(f * f);
    const size_t o_ntw = append(f * f * t);
    const size_t o_ntb = append(t * 2 * f);
    const size_t o_ntbias = append(t);
    const size_t o_fc1w = append(bn * t);
    const size_t o_fc1b = append(bn);
    const size_t o_fc2w = append(bn);
    const size_t o_fc2b = append(1);
    while (packed.size() % 4) packed.push_back(0.f);
    const size_t o_ntwt = packed.size();      // NTN weight re-ordered [i][t][j] for the all-pairs prep kernel
    packed.resize(packed.size() + f * f * t);
    for (size_t i = 0; i < f; ++i)
        for (size_t tt = 0; tt < t; ++tt)
            for (size_t j = 0; j < f; ++j) packed[o_ntwt + (i * t + tt) * f + j] = packed[o_ntw + i * (f * t) + j * t + tt];
    // ---- bf16 three-plane copies of the seven folded weight matrices, in MFMA operand order (sgpr_internal.hpp)
    size_t off_wb[7];
    for (int b = 0; b < 7; ++b) {
        const int rows = b < 6 ? 2 * bs[b].cout : bs[b].cout, kp = kp_of[b];
        const int nks = kp == 64 ? 2 : 1, nct = rows / 16;
        std::vector<unsigned short> wb((size_t)nct * nks * 3 * 512, 0);
        const float* wf = packed.data() + off_wf[b];
        for (int ct = 0; ct < nct; ++ct)
            for (int st = 0; st < nks; ++st)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int l15 = lane & 15, lq = lane >> 4;
                        int kk;
                        if (nks == 2) kk = 32 * st + 8 * lq + j;
                        else kk = j < 4 ? 4 * lq + j : -1;
                        unsigned short pl[3] = {0, 0, 0};
                        if (kk >= 0 && kk < kp) split3_host(wf[(size_t)(ct * 16 + l15) * kp + kk], pl);
                        for (int q = 0; q < 3; ++q) wb[(((size_t)(ct * nks + st) * 3 + q) * 64 + lane) * 8 + j] = pl[q];
                    }
        off_wb[b] = packed.size();
        packed.resize(packed.size() + wb.size() / 2);
        memcpy(packed.data() + off_wb[b], wb.data(), wb.size() * sizeof(unsigned short));
    }

    // ---- the same seven matrices as two f16 planes (w = hi + lo), [column tile][k-step][plane][lane][8]
    size_t off_wh[7];
    bool f16_ok = true;
    for (int b = 0; b < 7; ++b) {
        const int rows = b < 6 ? 2 * bs[b].cout : bs[b].cout, kp = kp_of[b];
        const int nks = kp == 64 ? 2 : 1, nct = rows / 16;
        std::vector<unsigned short> wh((size_t)nct * nks * 2 * 512, 0);
        const float* wf = packed.data() + off_wf[b];
        for (int ct = 0; ct < nct; ++ct)
            for (int st = 0; st < nks; ++st)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int l15 = lane & 15, lq = lane >> 4;
                        int kk;
                        if (nks == 2) kk = 32 * st + 8 * lq + j;
                        else kk = j < 4 ? 4 * lq + j : -1;
                        unsigned short pl[2] = {0, 0};
                        if (kk >= 0 && kk < kp) {
                            const float v = wf[(size_t)(ct * 16 + l15) * kp + kk];
                            if (!(fabsf(v) < 60000.f)) f16_ok = false;
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)(v - (float)hi);
                            memcpy(&pl[0], &hi, 2);
                            memcpy(&pl[1], &lo, 2);
                        }
                        for (int q = 0; q < 2; ++q) wh[(((size_t)(ct * nks + st) * 2 + q) * 64 + lane) * 8 + j] = pl[q];
                    }
        off_wh[b] = packed.size();
        packed.resize(packed.size() + wh.size() / 2);
        memcpy(packed.data() + off_wh[b], wh.data(), wh.size() * sizeof(unsigned short));
    }

    while (packed.size() % 4) packed.push_back(0.f);
    const size_t o_semtab = packed.size();      // graph-independent tables of the super-node branch + the largest value in them
    packed.resize(packed.size() + kSemTableFloats + 4, 0.f);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
    if (device < 0 || device >= ndev) {
        set_error("sgpr_create: device " + std::to_string(device) + " of " + std::to_string(ndev));
        return SGPR_E_INVALID;
    }
    DeviceGuard guard(device);   // the caller's current device is restored on return
    sgpr_handle* h = new sgpr_handle();
    memset(h, 0, sizeof(*h));
    h->device = device;
    h->dims = user_dims;            // (what the caller loaded; the kernels run on the built shapes)
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        h->num_cus = cus;
    }
    h->blob_floats = packed.size();
    e = hipMalloc(reinterpret_cast<void**>(&h->d_blob), packed.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_status), sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(h->d_blob, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(h->d_status, 0, sizeof(int32_t));
    if (e != hipSuccess) {
        if (h->d_blob) (void)hipFree(h->d_blob);
        if (h->d_status) (void)hipFree(h->d_status);
        delete h;
        return hip_fail(e, "sgpr_create: device allocation / upload");
    }
    // blob order is s1,f1,s2,f2,s3,f3 ; kernel order is f1,f2,f3,s1,s2,s3 (the semantic branch runs first: its 12
    // input registers per thread die after the first staging, the 3 xyz registers wait for the second)
    static const int blob_of_layer[6] = {1, 3, 5, 0, 2, 4};
    for (int l = 0; l < 6; ++l) {
        const int b = blob_of_layer[l];
        h->w.wf[l] = h->d_blob + off_wf[b];
        h->w.tb[l] = h->d_blob + off_tb[b];
        h->w.wb[l] = reinterpret_cast<const unsigned short*>(h->d_blob + off_wb[b]);
        h->w.wh[l] = reinterpret_cast<const unsigned short*>(h->d_blob + off_wh[b]);
        h->w.kp[l] = kp_of[b];
        h->w.cout[l] = bs[b].cout;
    }
    h->w.wf_end = h->d_blob + off_wf[6];
    h->w.tb_end = h->d_blob + off_tb[6];
    h->w.wb_end = reinterpret_cast<const unsigned short*>(h->d_blob + off_wb[6]);
    h->w.wh_end = reinterpret_cast<const unsigned short*>(h->d_blob + off_wh[6]);
    h->f16_weights = f16_ok ? 1 : 0;
    h->w.att_w = h->d_blob + o_att;
    h->w.ntn_w = h->d_blob + o_ntw;
    h->w.ntn_wt = h->d_blob + o_ntwt;
    h->w.ntn_wb = h->d_blob + o_ntb;
    h->w.ntn_bias = h->d_blob + o_ntbias;
    h->w.fc1_w = h->d_blob + o_fc1w;
    h->w.fc1_b = h->d_blob + o_fc1b;
    h->w.fc2_w = h->d_blob + o_fc2w;
    h->w.fc2_b = h->d_blob + o_fc2b;
    h->w.sem_g = h->w.sem_xx = h->w.sem_a2 = h->w.sem_b2 = nullptr;
    if (f16_ok) {
        // the graph-independent part of the super-node branch, by the kernels' own instructions (bit-identical to the
        // per-graph path); skipped when a layer-1 output leaves the f16 range (the per-graph path then flags the graph)
        float* tab = h->d_blob + o_semtab;
        float vmax = INFINITY;
        int rc = launch_sem_tables(h->w, tab, tab + kSemTableFloats, nullptr);
        if (rc == SGPR_OK) {
            e = hipMemcpy(&vmax, tab + kSemTableFloats, sizeof(float), hipMemcpyDeviceToHost);   // (synchronises)
            if (e != hipSuccess) rc = hip_fail(e, "sgpr_create: super-node tables");
        }
        if (rc != SGPR_OK) {
            (void)hipFree(h->d_blob);
            (void)hipFree(h->d_status);
            delete h;
            return rc;
        }
        if (vmax < 60000.f) {
            h->w.sem_g = tab;
            h->w.sem_xx = tab + 4 * 256;
            h->w.sem_a2 = h->w.sem_xx + 32;
            h->w.sem_b2 = h->w.sem_a2 + 2 * 16 * 64;
        }
    }
    // the any-shape model of the same checkpoint: serves node_num / K beyond the tuned kernels' limits
    h->generic_only = 0;
    h->d_gblob = nullptr;
    {
        const int rc = build_generic_model(user_weights, &user_dims, h);
        if (rc != SGPR_OK) {
            (void)hipFree(h->d_blob);
            (void)hipFree(h->d_status);
            if (h->d_gblob) (void)hipFree(h->d_gblob);      // (allocated, then its upload failed)
            delete h;
            return rc;
        }
    }
    *out = h;
    return SGPR_OK;
}

void sgpr_destroy(sgpr_handle* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    if (h->d_gblob) (void)hipFree(h->d_gblob);
    if (h->d_wblob) (void)hipFree(h->d_wblob);
    if (h->d_blob) (void)hipFree(h->d_blob);
    if (h->d_status) (void)hipFree(h->d_status);
    delete h;
}

#ifndef SGPR_SPLIT_SEM
#define SGPR_SPLIT_SEM 1     // lean production launches on packed input hand the semantic branch to semantic waves
#endif

// wide: use the wide-range X layouts (bf16 planes / fp32 rows) instead of the default f16 planes
// debug-mask bits that keep the production kernel instance: 8192 forces the wide-range instance, 1 << 20 makes the split
// launch's odd producers withhold their flag (test hook for the late-producer hand-over to the second pass)
static constexpr int kProductionSkipBits = 8192 | (1 << 20);
#ifndef SGPR_AUTO_LEAN
#define SGPR_AUTO_LEAN 1      // 0 (A/B builds): a launch without a node_cap promise is sized for node_num
#endif
static bool wide_range(const sgpr_handle* h) { return !h->f16_weights || (h->dbg_skip & 8192); }

static int check_nk(int G, int N, int k, int node_cap, EmbedPlan* plan, bool wide, bool small_park = false) {
    if (G < 0) {
        set_error("negative graph count");
        return SGPR_E_INVALID;
    }
    if (N < 1 || N > SGPR_MAX_NODES) {
        set_error("node_num " + std::to_string(N) + " outside [1, " + std::to_string(SGPR_MAX_NODES) + "]");
        return SGPR_E_NODES;
    }
    if (k < 1 || k > SGPR_MAX_K || k > N) {
        set_error("K " + std::to_string(k) + " outside [1, min(node_num, " + std::to_string(SGPR_MAX_K) + ")]");
        return SGPR_E_K;
    }
    if (node_cap < 0) {
        set_error("negative node_cap");
        return SGPR_E_INVALID;
    }
    if (!make_embed_plan(N, node_cap, k, plan, wide, small_park)) {
        set_error("no LDS plan for node_num " + std::to_string(N) + ", K " + std::to_string(k));
        return SGPR_E_NODES;
    }
    return SGPR_OK;
}

// workspace of an embed launch over G graphs of N slots:  redo flags [G] (one byte per launch slot, written by the f16
// instance, read by the wide-range second pass) | parked first-branch output [G][round16(N)][32] f32 when N > 128
// (sized for the uncapped plan: the second pass never uses a node_cap)
// (+ one word at the end of the flag region: EmbedArgs::redo_count)
static size_t embed_flag_bytes(int G) { return ((size_t)G + 8 + 255) & ~(size_t)255; }
static unsigned* embed_redo_count(unsigned char* flags, int G) {
    return reinterpret_cast<unsigned*>(flags + embed_flag_bytes(G) - 8);
}
static size_t embed_park_bytes(int G, int N) {
    return ((N > 128 ? (size_t)G * ((N + 15) / 16 * 16) * 32 * sizeof(float) : 0) + 255) & ~(size_t)255;
}
// ... | split launch (sgpr_internal.hpp, EmbedArgs::sem_tab): one 64-bit flag (the flag array rounded up to 16 bytes: the
// rows behind it are read and written as float4) + 16 rows of 32 floats per launch slot; a split launch has at most
// num_cus / 2 graphs (launch_embed), so the region is sized for that many slots, not for G
// The split launch serves at most min(num_cus / 2, kMaxSplitGraphs) graphs: launch_embed needs a CU per workgroup of
// both halves, and the workspace region is sized for kMaxSplitGraphs slots whatever the device (workspace queries work
// without a device: MI355X's 256 CUs / 2).  On a part with more CUs, launches of 129 .. num_cus / 2 graphs run unsplit
// (same results; the latency form simply stops at 128 graphs).
constexpr int kMaxSplitGraphs = 128;
static int embed_sem_slots(const sgpr_handle*, int G) { return G < kMaxSplitGraphs ? G : kMaxSplitGraphs; }
static size_t embed_sem_flag_bytes(int slots) { return ((size_t)slots * sizeof(unsigned long long) + 15) & ~(size_t)15; }
static size_t embed_sem_bytes(const sgpr_handle* h, int G) {
    const int slots = embed_sem_slots(h, G);
    return embed_sem_flag_bytes(slots) + (size_t)slots * 16 * 32 * sizeof(float);
}
static size_t embed_ws_bytes(const sgpr_handle* h, int G, int N) {
    return embed_flag_bytes(G) + embed_park_bytes(G, N) + embed_sem_bytes(h, G);
}

// which launches run on the any-shape kernels (sgpr_generic.hip): every launch of a handle whose architecture is larger
// than the built shape, and launches beyond the tuned kernels' node_num / K limits on any handle
static bool needs_generic(const sgpr_handle* h, int N, int k) {
    return h->generic_only || N > SGPR_MAX_NODES || k > SGPR_MAX_K;
}
static bool generic_nk_ok(int N, int k) {
    return N >= 1 && N <= SGPR_GENERIC_MAX_NODES && k >= 1 && k <= SGPR_GENERIC_MAX_K && k <= N;
}
static int pooled_width(const sgpr_handle* h) { return h->generic_only ? h->gm.f3 : kF3; }

// the matrix-core any-shape embed (sgpr_wide.hip) flags the graphs it hands to the plain-fp32 kernel: one byte per launch slot
// ahead of that kernel's scratch area
static size_t wide_flag_bytes(const sgpr_handle* h, int G, int N, int k) {
    return (h->generic_only && h->wm.ok && k == 10 && N <= SGPR_WIDE_MAX_NODES) ? (((size_t)G + 8 + 255) & ~(size_t)255) : 0;   // (+ the token word)
}

size_t sgpr_embed_workspace_bytes(const sgpr_handle* h, int G, int N, int k) {
    EmbedPlan p;
    if (!h || G < 0) return 0;
    if (needs_generic(h, N, k))
        return generic_nk_ok(N, k) ? generic_embed_ws_bytes(h, G, N, k) + wide_flag_bytes(h, G, N, k) + 256 : 0;
    if (!make_embed_plan(N, 0, k, &p)) return 0;
    return embed_ws_bytes(h, G, N);
}

size_t sgpr_embed_lds_bytes(const sgpr_handle* h, int N, int k) {
    EmbedPlan p;
    if (!h) return 0;
    if (needs_generic(h, N, k)) return generic_nk_ok(N, k) ? generic_embed_lds_bytes(h, N, k) : 0;
    if (!make_embed_plan(N, 0, k, &p)) return 0;
    return (size_t)p.lds_bytes;
}

int sgpr_pooled_width(const sgpr_handle* h) { return h ? pooled_width(h) : 0; }
int sgpr_is_any_shape(const sgpr_handle* h) { return (h && h->generic_only) ? 1 : 0; }

// an embed launch on the any-shape kernels
static int embed_generic(const sgpr_handle* h, EmbedArgs a, int N, int k, void* ws, size_t ws_bytes, void* stream) {
    if ((a.dbg_layers || a.dbg_knn) && h->generic_only) {
        set_error("sgpr_embed_debug: the layer dump's rows are 64 floats - not served for an architecture beyond the built shape");
        return SGPR_E_DIMS;
    }
    if (a.G < 0) {
        set_error("negative graph count");
        return SGPR_E_INVALID;
    }
    if (N < 1 || N > SGPR_GENERIC_MAX_NODES) {
        set_error("node_num " + std::to_string(N) + " outside [1, " + std::to_string(SGPR_GENERIC_MAX_NODES) + "]");
        return SGPR_E_NODES;
    }
    if (k < 1 || k > SGPR_GENERIC_MAX_K || k > N) {
        set_error("K " + std::to_string(k) + " outside [1, min(node_num, " + std::to_string(SGPR_GENERIC_MAX_K) + ")]");
        return SGPR_E_K;
    }
    const size_t flags = wide_flag_bytes(h, a.G, N, k);
    const size_t need = generic_embed_ws_bytes(h, a.G, N, k) + flags;
    if (need > 0 && (!ws || ws_bytes < need)) {
        set_error("sgpr_embed: workspace of " + std::to_string(need) + " bytes required (sgpr_embed_workspace_bytes)");
        return SGPR_E_WORKSPACE;
    }
    a.status = h->d_status;
    a.num_labels = h->dims.num_labels;
    a.skip = h->dbg_skip;                // (bits 24..27: ablation of the any-shape kernel's phases, tools/run_anyshape.py)
    DeviceGuard guard(h->device);
    // a moderately larger architecture (labels <= 32, filters <= 128 / 128 / 64) at node_num <= 112, K = 10: the matrix-core
    // embed; the plain-fp32 kernel then takes only the graphs it flagged (values outside the f16 range)
    if (flags > 0 && wide_embed_serves(h, a, N, k)) {
        a.redo = static_cast<unsigned char*>(ws);
        a.redo_count = reinterpret_cast<unsigned*>(a.redo + flags - 8);
        {
            // this call's token: a counter spread over 32 bits (a fresh workspace holds stale small integers - labels, flags,
            // earlier counters - which a bare counter meets by chance; a match costs the pass a scan of the flags, never a result)
            static std::atomic<unsigned> epoch{0u};
            unsigned e = (++epoch) * 0x9E3779B1u;
            if (e == 0u) e = 0x9E3779B1u;
            a.sem_epoch = e;
        }
        const int rc = launch_embed_wide(h, a, N, k, static_cast<hipStream_t>(stream));
        if (rc != SGPR_OK) return rc;
        a.auto_over = 7;
    }
    return launch_embed_generic(h, a, N, k, static_cast<unsigned char*>(ws) + flags, static_cast<hipStream_t>(stream));
}

static int embed_common(const sgpr_handle* h, EmbedArgs a, int N, int k, int node_cap, void* ws, size_t ws_bytes,
                        void* stream, int total_graphs = -1) {   // total_graphs: G when a.ids lists a subset
    if (h && a.G == 0) return SGPR_OK;
    if (!h || !a.pooled || (!a.dense && (!a.centers || (!a.labels && !(a.rag_off && a.rag_lab))))) {
        set_error("sgpr_embed: NULL argument");
        return SGPR_E_INVALID;
    }
    if (needs_generic(h, N, k)) return embed_generic(h, a, N, k, ws, ws_bytes, stream);
    // with at most one graph per CU there is nothing to overlap: keep the 512-thread workgroups (lower latency)
    a.promise = (node_cap > 0 && node_cap < N) ? node_cap : N;   // still enforced (a broken promise stays loud)
    const bool promised = node_cap > 0 && node_cap < N;
    if (a.G <= h->num_cus) node_cap = 0;
    EmbedPlan plan;
    // production launches (no dumps, timers or ablation) of lean plans park only the super-node rows
    const bool production = !a.dbg_layers && !a.dbg_knn && !h->dbg_prof && !(h->dbg_skip & ~kProductionSkipBits);
    // No promise, more graphs than CUs (the throughput regime): the launch still runs on the lean 64-row plan - four
    // workgroups per CU, what KITTI-like data fits into - and a graph with more processed slots is handed to the
    // owned-rows instance sized for node_num in the same call (launch_embed), instead of every graph paying for the
    // largest one could be.  (The reference has no such prerequisite either: sg_net.py:503-525.)
    // A promise of 65 .. kTwoTierCap slots (K <= 16) takes the same two tiers: the graphs of up to 64 slots on the lean plan,
    // the others on the owned-rows instance sized for the PROMISED cap - a mixed data set no longer pays every graph at the
    // size of its largest; the promise stays enforced (a graph beyond it: NaN + SGPR_E_NODES).  Measured (tools/run_auto.py,
    // ordered launches, 4541 graphs of node_num 100): 25..70 nodes (cap 71, 15 % above 64 slots) 322 -> 207 us, 40..85
    // nodes (cap 86, 48 %) 352 -> 334; but 2048 graphs of 20..120 nodes in 256 slots (cap 121, 56 % above 64 and most of the
    // work in them) 153 -> 188: two launches that each under-fill the device.  The library sees the cap, not the data: the
    // two tiers stop at a cap of 96 (six row tiles), where an oversize graph costs little more than a lean one.
    constexpr int kTwoTierCap = 96;
    bool auto_lean = false;
    const int over_cap = promised ? node_cap : N;
    if ((!promised || (node_cap > 64 && node_cap <= kTwoTierCap)) && production && !wide_range(h) && a.G > h->num_cus && N > 64 &&
        SGPR_AUTO_LEAN) {
        EmbedPlan lean;
        if (make_embed_plan(N, 64, k, &lean, false, true) && lean.lean) {
            node_cap = 64;
            auto_lean = true;
        }
    }
    int rc = check_nk(a.G, N, k, node_cap, &plan, wide_range(h), production);
    if (rc != SGPR_OK) return rc;
    // graphs are addressed by their own index: an ordered launch needs rows for all of them
    const int gtot = total_graphs < 0 ? a.G : total_graphs;
    const size_t need = embed_ws_bytes(h, gtot, N);
    if (!ws || ws_bytes < need) {
        set_error("sgpr_embed: workspace of " + std::to_string(need) + " bytes required (sgpr_embed_workspace_bytes)");
        return SGPR_E_WORKSPACE;
    }
    a.redo = static_cast<unsigned char*>(ws);
    a.redo_count = embed_redo_count(a.redo, gtot);
    a.over_count = a.redo_count + 1;                 // (the second word of the 8 bytes behind the flags)
    a.auto_over = auto_lean ? 1 : 0;
    a.over_cap = over_cap;
    a.park_ws = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + embed_flag_bytes(gtot));
    unsigned char* sem = static_cast<unsigned char*>(ws) + embed_flag_bytes(gtot) + embed_park_bytes(gtot, N);
    a.sem_flag = reinterpret_cast<unsigned long long*>(sem);               // indexed by launch slot (< a.G <= num_cus / 2)
    a.sem_tab = reinterpret_cast<float*>(sem + embed_sem_flag_bytes(embed_sem_slots(h, gtot)));
    if (a.G > kMaxSplitGraphs) a.sem_tab = nullptr;                        // (no split launch: the region holds fewer than G slots)
#if !SGPR_SPLIT_SEM
    a.sem_tab = nullptr;                                                   // (A/B builds: the unsplit launch)
#endif
    a.status = h->d_status;
    a.prof = h->dbg_prof;
    a.skip = h->dbg_skip;
    a.num_labels = h->dims.num_labels;
    DeviceGuard guard(h->device);
    return launch_embed(h, plan, a, static_cast<hipStream_t>(stream));
}

int sgpr_embed(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, int G, int N, int k,
               float* d_pooled, float* d_att, float* d_emb, void* d_workspace, size_t workspace_bytes,
               void* stream) {
    EmbedArgs a;
    memset(&a, 0, sizeof(a));
    a.centers = d_centers;
    a.labels = d_labels;
    a.G = G;
    a.pooled = d_pooled;
    a.att = d_att;
    a.emb = d_emb;
    return embed_common(h, a, N, k, 0, d_workspace, workspace_bytes, stream);
}

int sgpr_embed_capped(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, int G, int N, int node_cap,
                      int k, float* d_pooled, float* d_att, float* d_emb, void* d_workspace, size_t workspace_bytes,
                      void* stream) {
    EmbedArgs a;
    memset(&a, 0, sizeof(a));
    a.centers = d_centers;
    a.labels = d_labels;
    a.G = G;
    a.pooled = d_pooled;
    a.att = d_att;
    a.emb = d_emb;
    return embed_common(h, a, N, k, node_cap, d_workspace, workspace_bytes, stream);
}

int sgpr_embed_ordered(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, int G, int N, int node_cap,
                       int k, const int32_t* d_order, int n_order, float* d_pooled, float* d_att, float* d_emb,
                       void* d_workspace, size_t workspace_bytes, void* stream) {
    if (G < 0 || n_order < 0 || n_order > G || (n_order > 0 && !d_order)) {
        set_error("sgpr_embed_ordered: order list of " + std::to_string(n_order) + " entries for " + std::to_string(G) +
                  " graphs");
        return SGPR_E_INVALID;
    }
    EmbedArgs a;
    memset(&a, 0, sizeof(a));
    a.centers = d_centers;
    a.labels = d_labels;
    a.ids = d_order;
    a.G = n_order;
    a.pooled = d_pooled;
    a.att = d_att;
    a.emb = d_emb;
    return embed_common(h, a, N, k, node_cap, d_workspace, workspace_bytes, stream, G);
}

size_t sgpr_size_order_workspace_bytes(int G) { return G < 0 ? 0 : size_order_ws_bytes(G); }

int sgpr_size_order(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, const int64_t* d_offsets, int G,
                    int N, int k, int32_t* d_order, int32_t* d_info, void* d_workspace, size_t workspace_bytes,
                    void* stream) {
    if (!h || G < 0 || !d_info || (G > 0 && (!d_order || (!d_offsets && (!d_centers || !d_labels))))) {
        set_error("sgpr_size_order: NULL argument or negative count");
        return SGPR_E_INVALID;
    }
    if (N < 1 || N > SGPR_MAX_NODES || k < 1 || k > N) {
        set_error("sgpr_size_order: node_num " + std::to_string(N) + " / K " + std::to_string(k) +
                  " outside the tuned kernels' range (the any-shape kernels take no node_cap)");
        return SGPR_E_NODES;
    }
    if (!d_workspace || workspace_bytes < size_order_ws_bytes(G)) {
        set_error("sgpr_size_order: workspace of " + std::to_string(size_order_ws_bytes(G)) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    DeviceGuard guard(h->device);
    return launch_size_order(d_centers, d_labels, reinterpret_cast<const long long*>(d_offsets), G, N, k, h->dims.num_labels,
                             d_order, d_info, d_workspace, static_cast<hipStream_t>(stream));
}

int sgpr_embed_ragged(const sgpr_handle* h, const float* d_centers, const int8_t* d_labels, const int64_t* d_offsets,
                      int G, int N, int node_cap, int k, const int32_t* d_order, int n_order, float* d_pooled,
                      float* d_att, float* d_emb, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (G < 0 || n_order < 0 || n_order > G || (n_order > 0 && !d_order) || (G > 0 && !d_offsets)) {
        set_error("sgpr_embed_ragged: order list of " + std::to_string(n_order) + " entries for " + std::to_string(G) +
                  " graphs, or no offsets");
        return SGPR_E_INVALID;
    }
    EmbedArgs a;
    memset(&a, 0, sizeof(a));
    a.centers = d_centers;
    a.rag_lab = reinterpret_cast<const signed char*>(d_labels);
    a.rag_off = reinterpret_cast<const long long*>(d_offsets);
    a.ids = d_order;
    a.G = d_order ? n_order : G;
    a.pooled = d_pooled;
    a.att = d_att;
    a.emb = d_emb;
    return embed_common(h, a, N, k, node_cap, d_workspace, workspace_bytes, stream, G);
}

int sgpr_embed_dense(const sgpr_handle* h, const float* d_features, int G, int N, int k, float* d_pooled,
                     float* d_att, float* d_emb, void* d_workspace, size_t workspace_bytes, void* stream) {
    EmbedArgs a;
    memset(&a, 0, sizeof(a));
    a.dense = d_features;
    a.G = G;
    a.pooled = d_pooled;
    a.att = d_att;
    a.emb = d_emb;
    return embed_common(h, a, N, k, 0, d_workspace, workspace_bytes, stream);
}

int sgpr_embed_debug(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, int G, int N, int k,
                     float* d_pooled, float* d_att, float* d_emb, float* d_layers, int32_t* d_knn,
                     void* d_workspace, size_t workspace_bytes, void* stream) {
    EmbedArgs a;
    memset(&a, 0, sizeof(a));
    a.centers = d_centers;
    a.labels = d_labels;
    a.G = G;
    a.pooled = d_pooled;
    a.att = d_att;
    a.emb = d_emb;
    a.dbg_layers = d_layers;
    a.dbg_knn = d_knn;
    return embed_common(h, a, N, k, 0, d_workspace, workspace_bytes, stream);
}

int sgpr_score_pairs(const sgpr_handle* h, const float* d_pooled1, const int32_t* d_idx1, const float* d_pooled2,
                     const int32_t* d_idx2, int64_t P, float* d_score, void* stream) {
    if (!h || !d_pooled1 || !d_pooled2 || !d_score || P < 0) {
        set_error("sgpr_score_pairs: NULL argument or negative count");
        return SGPR_E_INVALID;
    }
    DeviceGuard guard(h->device);
    if (h->generic_only)
        return launch_score_generic(h, d_pooled1, d_idx1, d_pooled2, d_idx2, P, 0, d_score, 0, static_cast<hipStream_t>(stream));
    return launch_score_pairs(h, d_pooled1, d_idx1, d_pooled2, d_idx2, P, d_score, static_cast<hipStream_t>(stream));
}

size_t sgpr_pair_plan_ints(int64_t P, int R) {
    if (P < 0 || R < 0) return 0;
    const int64_t rows = P < R ? P : R;                       // distinct row graphs at most
    const int64_t items = P / 16 + rows;                      // sum over rows of ceil(count / 16) at most
    return (size_t)(rows + items + (items + 1) + 2 * P);
}

int sgpr_pair_plan(const int32_t* h_idx1, const int32_t* h_idx2, int64_t P, int R, int M, int32_t* h_plan,
                   size_t plan_capacity_ints, size_t* plan_ints, int32_t* n_rows, int32_t* n_items) {
    if (P < 0 || P >= 0x7fffffffLL || R < 0 || M < 0 || (P > 0 && (!h_idx1 || !h_idx2)) || !plan_ints || !n_rows || !n_items) {
        set_error("sgpr_pair_plan: NULL argument, negative count or 2^31 pairs or more");
        return SGPR_E_INVALID;
    }
    // (host allocations proportional to R: a failure is an error code, never an exception across the C boundary)
    std::vector<int32_t> count, cursor;
    try {
        count.assign((size_t)R + 1, 0);
        cursor.assign((size_t)R + 1, 0);
    } catch (const std::exception&) {
        set_error("sgpr_pair_plan: out of host memory for " + std::to_string(R) + " row graphs");
        return SGPR_E_INVALID;
    }
    for (int64_t p = 0; p < P; ++p) {
        const int32_t a = h_idx1[p], b = h_idx2[p];
        if (a < 0 || a >= R || b < 0 || b >= M) {
            set_error("sgpr_pair_plan: pair " + std::to_string(p) + " = (" + std::to_string(a) + ", " + std::to_string(b) +
                      ") outside [0, " + std::to_string(R) + ") x [0, " + std::to_string(M) + ")");
            return SGPR_E_INVALID;
        }
        ++count[a];
    }
    int64_t nr = 0, ni = 0;
    for (int r = 0; r < R; ++r)
        if (count[r]) {
            ++nr;
            ni += (count[r] + 15) / 16;
        }
    const size_t need = (size_t)(nr + ni + (ni + 1) + 2 * P);
    *plan_ints = need;
    *n_rows = (int32_t)nr;
    *n_items = (int32_t)ni;
    if (!h_plan || plan_capacity_ints < need) {
        if (!h_plan && plan_capacity_ints == 0) return SGPR_OK;           // size query
        set_error("sgpr_pair_plan: plan needs " + std::to_string(need) + " int32 words");
        return SGPR_E_WORKSPACE;
    }
    int32_t* row_ids = h_plan;
    int32_t* item_row = row_ids + nr;
    int32_t* item_beg = item_row + ni;
    int32_t* cols = item_beg + ni + 1;
    int32_t* pos = cols + P;
    int32_t at = 0, cr = 0, it = 0;
    for (int r = 0; r < R; ++r) {
        cursor[r] = at;
        if (!count[r]) continue;
        row_ids[cr] = r;
        for (int32_t c0 = 0; c0 < count[r]; c0 += 16) {
            item_row[it] = cr;
            item_beg[it++] = at + c0;
        }
        at += count[r];
        ++cr;
    }
    item_beg[it] = (int32_t)P;
    for (int64_t p = 0; p < P; ++p) {
        const int32_t w = cursor[h_idx1[p]]++;
        cols[w] = h_idx2[p];
        pos[w] = (int32_t)p;
    }
    return SGPR_OK;
}

size_t sgpr_score_pair_list_workspace_bytes(const sgpr_handle* h, int n_rows, int M) {
    if (!h || n_rows < 0 || M < 0) return 0;
    return score_pair_list_ws_bytes(n_rows, M);
}

int sgpr_score_pair_list(const sgpr_handle* h, const float* d_pooled_rows, int R, const float* d_pooled_cols, int M,
                         const int32_t* d_plan, int n_rows, int n_items, int64_t P, float* d_score, void* d_workspace,
                         size_t workspace_bytes, void* stream) {
    if (!h || R < 0 || M < 0 || P < 0 || P >= 0x7fffffffLL || n_rows < 0 || n_rows > R || n_items < 0 ||
        (P > 0 && (!d_pooled_rows || !d_pooled_cols || !d_plan || !d_score || n_rows == 0 || n_items == 0 || n_items > P))) {
        set_error("sgpr_score_pair_list: NULL argument, negative count or a plan that does not fit (P, R)");
        return SGPR_E_INVALID;
    }
    if (P == 0) return SGPR_OK;
    if (h->generic_only) {                                   // (an architecture beyond the built shape: the plan walked on the
        DeviceGuard guard(h->device);                        //  any-shape tail, pair by pair - the bits of sgpr_score_pairs; no workspace)
        return launch_score_plan_generic(h, d_pooled_rows, d_pooled_cols, d_plan, n_rows, n_items, P, d_score,
                                         static_cast<hipStream_t>(stream));
    }
    const size_t need = score_pair_list_ws_bytes(n_rows, M);
    if (!d_workspace || workspace_bytes < need) {
        set_error("sgpr_score_pair_list: workspace of " + std::to_string(need) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    DeviceGuard guard(h->device);
    return launch_score_pair_list(h, d_pooled_rows, d_pooled_cols, M, d_plan, n_rows, n_items, P, d_score, d_workspace,
                                  static_cast<hipStream_t>(stream));
}

size_t sgpr_score_all_pairs_workspace_bytes(const sgpr_handle* h, int R, int M) {
    if (!h || R < 0 || M < 0) return 0;
    if (h->generic_only && h->wm.ok) return std::max(score_all_pairs_ws_bytes(R, M), wide_tail_ws_bytes(R, M));
    return score_all_pairs_ws_bytes(R, M);
}

// One rectangle of an any-shape handle: a moderately larger tensor network (pooled width <= 64, <= 32 neurons) on the matrix
// cores (sgpr_wide.hip) when the caller brought its workspace; the plain-fp32 kernel behind it runs only if the inputs left
// the f16 range (a device word).  Handles beyond those limits: the plain-fp32 kernel alone, no workspace.
static int score_rect_any_shape(const sgpr_handle* h, const float* rows, int R, const float* cols, int M, float* score, int64_t ld,
                                void* ws, size_t ws_bytes, hipStream_t stream) {
    const unsigned* gate = nullptr;
    if (wide_tail_serves(h) && ws && ws_bytes >= wide_tail_ws_bytes(R, M)) {
        const int rc = launch_score_all_pairs_wide_any(h, rows, R, cols, M, score, ld, ws, &gate, stream);
        if (rc != SGPR_OK) return rc;
    }
    return launch_score_generic(h, rows, nullptr, cols, nullptr, (int64_t)R * M, M, score, ld, stream, gate);
}

int sgpr_score_all_pairs(const sgpr_handle* h, const float* d_pooled_rows, int R, const float* d_pooled_cols, int M,
                         float* d_score, int64_t ld, void* d_workspace, size_t workspace_bytes, void* stream) {
    // an empty rectangle (a rank whose row shard is empty: fewer graphs than ranks) needs no buffers at all
    if (!h || R < 0 || M < 0 || ld < M || (R > 0 && M > 0 && (!d_pooled_rows || !d_pooled_cols || !d_score))) {
        set_error("sgpr_score_all_pairs: NULL argument, negative count or ld < M");
        return SGPR_E_INVALID;
    }
    if (R == 0 || M == 0) return SGPR_OK;
    if (h->generic_only) {
        DeviceGuard guard(h->device);
        return score_rect_any_shape(h, d_pooled_rows, R, d_pooled_cols, M, d_score, ld, d_workspace, workspace_bytes,
                                    static_cast<hipStream_t>(stream));
    }
    const size_t need = score_all_pairs_ws_bytes(R, M);
    if (need > 0 && (!d_workspace || workspace_bytes < need)) {
        set_error("sgpr_score_all_pairs: workspace of " + std::to_string(need) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    DeviceGuard guard(h->device);
    // (debug bit 13 / weights outside the f16 range: the instance with three bf16 planes per operand, as for the embed)
    return launch_score_all_pairs(h, d_pooled_rows, R, d_pooled_cols, M, d_score, ld, d_workspace,
                                  static_cast<hipStream_t>(stream), wide_range(h));
}

static int check_jobs(const sgpr_handle* h, int n, const sgpr_pairs_job* jobs) {
    if (!h || n < 0 || n > SGPR_MAX_PAIR_JOBS || (n > 0 && !jobs)) {
        set_error("sgpr_score_all_pairs_multi: NULL argument or not 0.." + std::to_string(SGPR_MAX_PAIR_JOBS) + " jobs");
        return SGPR_E_INVALID;
    }
    for (int j = 0; j < n; ++j) {
        const sgpr_pairs_job& q = jobs[j];
        if (q.R < 0 || q.M < 0 || q.ld < q.M ||
            (q.R > 0 && q.M > 0 && (!q.d_pooled_rows || !q.d_pooled_cols || !q.d_score))) {
            set_error("sgpr_score_all_pairs_multi: job " + std::to_string(j) + ": NULL pointer, negative count or ld < M");
            return SGPR_E_INVALID;
        }
    }
    return SGPR_OK;
}

size_t sgpr_score_all_pairs_multi_workspace_bytes(const sgpr_handle* h, int n_jobs, const sgpr_pairs_job* jobs) {
    if (check_jobs(h, n_jobs, jobs) != SGPR_OK) return 0;
    size_t any = 0;
    if (h->generic_only && h->wm.ok)
        for (int j = 0; j < n_jobs; ++j) any = std::max(any, wide_tail_ws_bytes(jobs[j].R, jobs[j].M));
    return std::max(any, score_all_pairs_multi_ws_bytes(n_jobs, jobs));
}

int sgpr_score_all_pairs_multi(const sgpr_handle* h, int n_jobs, const sgpr_pairs_job* jobs, void* d_workspace,
                               size_t workspace_bytes, void* stream) {
    int rc = check_jobs(h, n_jobs, jobs);
    if (rc != SGPR_OK) return rc;
    if (h->generic_only) {
        DeviceGuard guard(h->device);
        for (int j = 0; j < n_jobs && rc == SGPR_OK; ++j)
            if (jobs[j].R > 0 && jobs[j].M > 0)
                rc = score_rect_any_shape(h, jobs[j].d_pooled_rows, jobs[j].R, jobs[j].d_pooled_cols, jobs[j].M, jobs[j].d_score,
                                          jobs[j].ld, d_workspace, workspace_bytes, static_cast<hipStream_t>(stream));   // (stream order
                                                                                     // lets the jobs share one workspace)
        return rc;
    }
    const size_t need = score_all_pairs_multi_ws_bytes(n_jobs, jobs);
    if (need > 0 && (!d_workspace || workspace_bytes < need)) {
        set_error("sgpr_score_all_pairs_multi: workspace of " + std::to_string(need) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    DeviceGuard guard(h->device);
    return launch_score_all_pairs_multi(h, n_jobs, jobs, d_workspace, static_cast<hipStream_t>(stream));
}

// workspace of sgpr_forward_dense: pooled [2B][32] | embed workspace for 2B graphs
size_t sgpr_forward_workspace_bytes(const sgpr_handle* h, int B, int N, int k) {
    EmbedPlan p;
    if (!h || B < 0) return 0;
    if (needs_generic(h, N, k)) {
        if (!generic_nk_ok(N, k)) return 0;
        const size_t pooled = ((size_t)2 * B * pooled_width(h) * sizeof(float) + 255) / 256 * 256;
        return pooled + generic_embed_ws_bytes(h, 2 * B, N, k) + 256;
    }
    if (!make_embed_plan(N, 0, k, &p)) return 0;
    return (size_t)2 * B * kF3 * sizeof(float) + embed_ws_bytes(h, 2 * B, N);
}

// sgpr_forward_dense on the any-shape kernels: the two sides embed as one launch of 2B graphs (one launch per side when
// the attention buffers are separate), then one wave per pair
static int forward_dense_generic(const sgpr_handle* h, const float* f1, const float* f2, int B, int N, int k,
                                 float* d_score, float* d_att1, float* d_att2, void* d_workspace,
                                 size_t workspace_bytes, void* stream) {
    const size_t need = sgpr_forward_workspace_bytes(h, B, N, k);
    if (need == 0) {
        set_error("sgpr_forward_dense: node_num " + std::to_string(N) + " / K " + std::to_string(k) + " outside the any-shape "
                  "limits (node_num <= " + std::to_string(SGPR_GENERIC_MAX_NODES) + ", K <= min(node_num, " +
                  std::to_string(SGPR_GENERIC_MAX_K) + "))");
        return (N < 1 || N > SGPR_GENERIC_MAX_NODES) ? SGPR_E_NODES : SGPR_E_K;
    }
    if (!d_workspace || workspace_bytes < need) {
        set_error("sgpr_forward_dense: workspace of " + std::to_string(need) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    const int pw = pooled_width(h);
    float* pooled = static_cast<float*>(d_workspace);
    const size_t pooled_bytes = ((size_t)2 * B * pw * sizeof(float) + 255) / 256 * 256;
    void* ws = static_cast<char*>(d_workspace) + pooled_bytes;
    const size_t ws_bytes = workspace_bytes - pooled_bytes;
    EmbedArgs a;
    memset(&a, 0, sizeof(a));
    a.dense = f1;
    a.dense2 = f2;
    a.g_split = B;
    a.G = 2 * B;
    a.pooled = pooled;
    int rc;
    if ((d_att1 && d_att2 && d_att2 == d_att1 + (size_t)B * N) || (!d_att1 && !d_att2)) {
        a.att = d_att1;
        rc = embed_generic(h, a, N, k, ws, ws_bytes, stream);
    } else {
        EmbedArgs a1 = a, a2 = a;
        a1.dense2 = nullptr; a1.G = B; a1.att = d_att1;
        a2.dense = f2; a2.dense2 = nullptr; a2.G = B; a2.att = d_att2;
        a2.pooled = pooled + (size_t)B * pw;
        rc = embed_generic(h, a1, N, k, ws, ws_bytes, stream);
        if (rc == SGPR_OK) rc = embed_generic(h, a2, N, k, ws, ws_bytes, stream);
    }
    if (rc != SGPR_OK) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->generic_only)
        return launch_score_generic(h, pooled, nullptr, pooled + (size_t)B * pw, nullptr, B, 0, d_score, 0, s);
    return launch_score_pairs(h, pooled, nullptr, pooled + (size_t)B * kF3, nullptr, B, d_score, s);
}

int sgpr_forward_dense(const sgpr_handle* h, const float* d_features_1, const float* d_features_2, int B, int N,
                       int k, float* d_score, float* d_att1, float* d_att2, void* d_workspace,
                       size_t workspace_bytes, void* stream) {
    if (!h || !d_features_1 || !d_features_2 || !d_score || B < 0) {
        set_error("sgpr_forward_dense: NULL argument or negative batch");
        return SGPR_E_INVALID;
    }
    if (needs_generic(h, N, k))
        return forward_dense_generic(h, d_features_1, d_features_2, B, N, k, d_score, d_att1, d_att2, d_workspace,
                                     workspace_bytes, stream);
    EmbedPlan plan;
    int rc = check_nk(2 * B, N, k, 0, &plan, wide_range(h), !h->dbg_prof && !(h->dbg_skip & ~kProductionSkipBits));
    if (rc != SGPR_OK) return rc;
    const size_t need = sgpr_forward_workspace_bytes(h, B, N, k);
    if (!d_workspace || workspace_bytes < need) {
        set_error("sgpr_forward_dense: workspace of " + std::to_string(need) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    float* pooled = static_cast<float*>(d_workspace);
    // both sides in ONE launch of 2B workgroups: graphs [0,B) = side 1, [B,2B) = side 2
    EmbedArgs a;
    memset(&a, 0, sizeof(a));
    a.dense = d_features_1;
    a.dense2 = d_features_2;
    a.g_split = B;
    a.G = 2 * B;
    a.pooled = pooled;
    a.redo = reinterpret_cast<unsigned char*>(pooled + (size_t)2 * B * kF3);
    a.redo_count = embed_redo_count(a.redo, 2 * B);      // (the per-side launches below share the word: tokens differ)
    a.park_ws = reinterpret_cast<float*>(a.redo + embed_flag_bytes(2 * B));
    a.status = h->d_status;
    a.prof = h->dbg_prof;
    a.skip = h->dbg_skip;
    a.num_labels = h->dims.num_labels;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d_att1 && d_att2 && d_att2 == d_att1 + (size_t)B * N) {
        a.att = d_att1;  // contiguous [2B, N] attention buffer
        rc = launch_embed(h, plan, a, s);
    } else if (!d_att1 && !d_att2) {
        rc = launch_embed(h, plan, a, s);
    } else {
        // separate attention buffers: one launch per side
        EmbedArgs a1 = a, a2 = a;
        a1.dense2 = nullptr; a1.G = B; a1.att = d_att1;
        a2.dense = d_features_2; a2.dense2 = nullptr; a2.G = B; a2.att = d_att2;
        a2.pooled = pooled + (size_t)B * kF3;
        a2.redo = a.redo + B;
        a2.park_ws = a.park_ws + (size_t)B * ((N + 15) / 16 * 16) * 32;
        rc = launch_embed(h, plan, a1, s);
        if (rc == SGPR_OK) rc = launch_embed(h, plan, a2, s);
    }
    if (rc != SGPR_OK) return rc;
    return launch_score_pairs(h, pooled, nullptr, pooled + (size_t)B * kF3, nullptr, B, d_score, s);
}

int sgpr_knn(const float* d_x, int B, int C, int N, int k, int64_t* d_idx, void* stream) {
    if (!d_x || !d_idx || B < 0 || C < 1) {
        set_error("sgpr_knn: NULL argument, negative batch or no channels");
        return SGPR_E_INVALID;
    }
    if (N < 1 || N > SGPR_ANY_MAX_NODES) {
        set_error("sgpr_knn: N " + std::to_string(N) + " outside [1, " + std::to_string(SGPR_ANY_MAX_NODES) + "]");
        return SGPR_E_NODES;
    }
    if (k < 1 || k > SGPR_ANY_MAX_K || k > N) {
        set_error("sgpr_knn: K " + std::to_string(k) + " outside [1, min(N, " + std::to_string(SGPR_ANY_MAX_K) + ")]");
        return SGPR_E_K;
    }
    if (N > SGPR_MAX_NODES || k > SGPR_MAX_K)                    // beyond the LDS-resident kernel: one wave per row
        return launch_knn_any(d_x, B, C, N, k, d_idx, static_cast<hipStream_t>(stream));
    return launch_knn(d_x, B, C, N, k, d_idx, static_cast<hipStream_t>(stream));
}

int sgpr_graph_feature(const float* d_x, const int64_t* d_idx, int B, int C, int N, int k, float* d_out, void* stream) {
    if (!d_x || !d_idx || !d_out || B < 0 || C < 1 || N < 1 || k < 1) {
        set_error("sgpr_graph_feature: NULL argument or non-positive size");
        return SGPR_E_INVALID;
    }
    return launch_graph_feature(d_x, d_idx, B, C, N, k, d_out, static_cast<hipStream_t>(stream));
}

int sgpr_attention_pool(const float* d_weight, const float* d_emb, int B, int N, float* d_rep, float* d_att,
                        void* stream) {
    if (!d_weight || !d_emb || !d_rep || B < 0 || N < 1) {
        set_error("sgpr_attention_pool: NULL argument, negative batch or no nodes");
        return SGPR_E_INVALID;
    }
    if ((size_t)N * sizeof(float) > 48 * 1024) {
        set_error("sgpr_attention_pool: more than 12288 nodes per graph");
        return SGPR_E_NODES;
    }
    return launch_attention_pool(d_weight, d_emb, B, N, d_rep, d_att, static_cast<hipStream_t>(stream));
}

int sgpr_attention_pool_any(const float* d_weight, const float* d_emb, int B, int N, int F, float* d_rep, float* d_att,
                            void* stream) {
    if (!d_weight || !d_emb || !d_rep || B < 0 || N < 1) {
        set_error("sgpr_attention_pool_any: NULL argument, negative batch or no nodes");
        return SGPR_E_INVALID;
    }
    if (F < 1 || F > SGPR_ANY_MAX_FILTERS_3) {
        set_error("sgpr_attention_pool_any: width " + std::to_string(F) + " outside [1, " +
                  std::to_string(SGPR_ANY_MAX_FILTERS_3) + "]");
        return SGPR_E_DIMS;
    }
    return launch_attention_any(d_weight, d_emb, B, N, F, d_rep, d_att, static_cast<hipStream_t>(stream));
}

int sgpr_ntn_any(const float* d_weight, const float* d_weight_block, const float* d_bias, const float* d_e1,
                 const float* d_e2, int64_t B, int F, int T, float* d_out, void* stream) {
    if (!d_weight || !d_weight_block || !d_bias || !d_e1 || !d_e2 || !d_out || B < 0) {
        set_error("sgpr_ntn_any: NULL argument or negative batch");
        return SGPR_E_INVALID;
    }
    if (F < 1 || F > SGPR_ANY_MAX_FILTERS_3 || T < 1 || T > SGPR_ANY_MAX_NEURONS) {
        set_error("sgpr_ntn_any: width " + std::to_string(F) + " / " + std::to_string(T) + " neurons outside [1, " +
                  std::to_string(SGPR_ANY_MAX_FILTERS_3) + "] / [1, " + std::to_string(SGPR_ANY_MAX_NEURONS) + "]");
        return SGPR_E_DIMS;
    }
    return launch_ntn_any(d_weight, d_weight_block, d_bias, d_e1, d_e2, B, F, T, d_out, static_cast<hipStream_t>(stream));
}

int sgpr_ntn(const float* d_weight, const float* d_weight_block, const float* d_bias, const float* d_e1,
             const float* d_e2, int64_t B, float* d_out, void* stream) {
    if (!d_weight || !d_weight_block || !d_bias || !d_e1 || !d_e2 || !d_out || B < 0) {
        set_error("sgpr_ntn: NULL argument or negative batch");
        return SGPR_E_INVALID;
    }
    return launch_ntn(d_weight, d_weight_block, d_bias, d_e1, d_e2, B, d_out, static_cast<hipStream_t>(stream));
}

size_t sgpr_cluster_workspace_bytes(int P) { return P < 0 ? 0 : cluster_ws_bytes(P); }

int sgpr_cluster_scan(const float* d_points, int point_stride, const uint32_t* d_labels, int P, int max_nodes,
                      double* d_centers, int32_t* d_node_labels, int32_t* d_node_sizes, int32_t* d_point_node,
                      int32_t* d_num_nodes, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (P < 0 || max_nodes < 0 || point_stride < 3 || !d_num_nodes || (P > 0 && (!d_points || !d_labels)) ||
        (max_nodes > 0 && (!d_centers || !d_node_labels || !d_node_sizes))) {
        set_error("sgpr_cluster_scan: NULL argument, negative count or point_stride < 3");
        return SGPR_E_INVALID;
    }
    const size_t need = cluster_ws_bytes(P);
    if (!d_workspace || workspace_bytes < need) {
        set_error("sgpr_cluster_scan: workspace of " + std::to_string(need) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    return launch_cluster_scan(d_points, point_stride, d_labels, P, max_nodes, d_centers, d_node_labels, d_node_sizes,
                               d_point_node, d_num_nodes, d_workspace, static_cast<hipStream_t>(stream));
}

int sgpr_graph_edges(const float* d_points, int point_stride, const int32_t* d_point_node, int P, int n,
                     const double* d_centers, double* d_min_dis, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (P < 0 || n < 0 || n > 8192 || point_stride < 3 || (n > 0 && (!d_points || !d_point_node || !d_centers || !d_min_dis))) {
        set_error("sgpr_graph_edges: NULL argument, negative count or point_stride < 3");
        return SGPR_E_INVALID;
    }
    if (n > 0 && (!d_workspace || workspace_bytes < (size_t)n * n * sizeof(int))) {
        set_error("sgpr_graph_edges: workspace of " + std::to_string((size_t)n * n * sizeof(int)) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    return launch_graph_edges(d_points, point_stride, d_point_node, P, n, d_centers, d_min_dis, d_workspace,
                              static_cast<hipStream_t>(stream));
}

void sgpr_debug_set_skip_mask(sgpr_handle* h, int mask) {
    if (h) h->dbg_skip = mask;
}

int sgpr_debug_uses_f16_planes(const sgpr_handle* h) { return (h && h->f16_weights) ? 1 : 0; }

void sgpr_debug_set_profile_buffer(sgpr_handle* h, void* d_counters) {
    if (h) h->dbg_prof = static_cast<unsigned long long*>(d_counters);
}

int sgpr_check_status(const sgpr_handle* h, void* stream) {
    if (!h) {
        set_error("sgpr_check_status: NULL handle");
        return SGPR_E_INVALID;
    }
    int32_t flag = 0;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(&flag, h->d_status, sizeof(flag), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(e, "sgpr_check_status");
    if (flag) {
        e = hipMemsetAsync(h->d_status, 0, sizeof(flag), s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return hip_fail(e, "sgpr_check_status: reset");
        if (flag & 4) {
            set_error("internal: a semantic wave of the split embed launch did not deliver (pooled vector set to NaN)");
            return SGPR_E_HIP;
        }
        if (flag & 8) {
            set_error("a graph of the ragged store holds more nodes than node_num slots, or its offsets decrease (its pooled vector is NaN)");
            return SGPR_E_NODES;
        }
        if (flag & 2) {
            set_error("a graph needed more slots than the node_cap passed to sgpr_embed_capped (its pooled vector is NaN)");
            return SGPR_E_NODES;
        }
        set_error("a node label outside [-1, num_labels) was seen by the embed kernel");
        return SGPR_E_LABEL;
    }
    return SGPR_OK;
}

}  // extern "C"
