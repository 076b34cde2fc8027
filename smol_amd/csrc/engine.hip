// engine.hip -- C-ABI (include/smolmc.h), host-side table preparation and the evaluation kernels.
#include <array>
#include <map>
#include <string>
#include <thread>

#include "smolmc_common.h"

thread_local std::string smolmc_g_err;

// ----------------------------------------------------------------------------
// reference-layout evaluation kernels (parity API + initial trace)
// ----------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < nw; ++i) s += sh[i];
    return s;
}

// Ensemble.compute_feature_vector for one occupancy per block
// (evaluator.pyx:121-209 x size; processor/ewald.py:128-145; ensemble.py:343-349)
// ce_only: the cluster features alone (lazy cluster features: the scalar features are carried by the kernels)
__global__ void __launch_bounds__(256) eval_full_kernel(const RefTables T, const uint8_t *occ_all,
                                                        double *out_all, const int ce_only, const int nocc, const int M) {
    __shared__ double sh[8];
    extern __shared__ __attribute__((aligned(16))) unsigned char eval_smem[];
    // Round 5 (this kernel is on the sampling path of handles with lazy cluster features): a block evaluates M <= 4
    // occupancies at once -- the index rows of an orbit, 15 MB for config 2, are read once for the M of them -- from
    // copies in LDS (M x Npad bytes of dynamic LDS; M = 0: one occupancy, read through the caches).
    const int m_eff = M > 0 ? M : 1;
    const int first = blockIdx.x * m_eff;
    const uint8_t *gocc = occ_all + (size_t)first * T.Npad;
    if (M > 0) {
        for (int m = 0; m < M; ++m) {
            const int q = first + m < nocc ? m : 0; // (blocks past the end repeat their first occupancy)
            const uint4 *src = (const uint4 *)(gocc + (size_t)q * T.Npad);
            uint4 *dst = (uint4 *)(eval_smem + (size_t)m * T.Npad);
            for (int i = threadIdx.x; i < T.Npad / 16; i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
    }
    const uint8_t *occ = M > 0 ? (const uint8_t *)eval_smem : gocc;
    const int ostr = M > 0 ? T.Npad : 0; // occupancy m at occ + m * ostr
    if (threadIdx.x < m_eff && first + (int)threadIdx.x < nocc)
        out_all[(size_t)(first + threadIdx.x) * T.F] = (T.corr_mode ? 1.0 : T.offset) * (double)T.P;
    for (int n = 0; n < T.n_orb; ++n) {
        const int I = T.orb_nsites[n], K = T.corr_mode ? T.orb_nfunc[n] : 1;
        const int Nt = T.orb_tensor_len[n];
        const int *st = T.tensor_indices + T.orb_stride_off[n];
        const int *ind = T.full_idx + T.full_off[n];
        const long long J = (T.full_off[n + 1] - T.full_off[n]) / I;
        const double *t0 = T.corr_mode ? T.corr_tensors + T.orb_ctensor_off[n] : T.interaction_tensors + T.orb_itensor_off[n];
        // the tensor index of a cluster once for up to four functions (the sums of a function are those of the
        // function-by-function loop: same clusters per thread, same order)
        for (int k0 = 0; k0 < K; k0 += 4) {
            double p[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int k = 0; k < 4; ++k) p[m][k] = 0.0;
            const int kn = K - k0 < 4 ? K - k0 : 4;
            for (long long j = threadIdx.x; j < J; j += blockDim.x) {
                int index[4] = {0, 0, 0, 0};
                for (int i = 0; i < I; ++i) {
                    const int x = ind[j * I + i], sti = st[i];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (m < m_eff) index[m] += sti * (int)occ[(size_t)m * ostr + x];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (m < m_eff)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < kn) p[m][k] += t0[(size_t)(k0 + k) * Nt + index[m]];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < m_eff)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < kn) {
                            const double v = block_sum(p[m][k], sh);
                            if (threadIdx.x == 0 && first + m < nocc) {
                                const int o = T.corr_mode ? T.orb_bit_id[n] + k0 + k : T.orb_id[n];
                                out_all[(size_t)(first + m) * T.F + o] = v / (double)J * (double)T.P;
                            }
                        }
        }
    }
    if (ce_only) return;
    // (the scalar features below: launches with one occupancy per block)
    double *out = out_all + (size_t)first * T.F;
    int f = T.Fce;
    if (T.has_ewald) {
        double s = 0;
        for (int a = 0; a < T.N; ++a) {
            const int ia = T.ew_inds[(size_t)a * T.ew_W + occ[a]];
            if (ia == -1) continue;
            const double *row = T.ew_M_rowmajor + (size_t)ia * T.ew_M;
            for (int b = threadIdx.x; b < T.N; b += blockDim.x) {
                const int ib = T.ew_inds[(size_t)b * T.ew_W + occ[b]];
                if (ib != -1) s += row[ib];
            }
        }
        s = block_sum(s, sh);
        if (threadIdx.x == 0) out[f] = s;
        f++;
    }
    if (T.has_mu) {
        double s = 0;
        for (int a = threadIdx.x; a < T.N; a += blockDim.x) s += T.mu[(size_t)a * T.mu_W + occ[a]];
        s = block_sum(s, sh);
        if (threadIdx.x == 0) out[f] = s;
    }
}

// Ensemble.compute_feature_vector_change for one step (<= SMOLMC_MAX_STEP_FLIPS sequential flips) per
// wave, reference table layout and arithmetic chain p / ratio / J, x size
// (evaluator.pyx:244-262, :302-315; expansion.py:217-231).  Flip f sees the flips before it: a site's
// species is the code of the LAST earlier flip there, else the occupancy's.
__global__ void __launch_bounds__(64) eval_delta_kernel(const RefTables T, const uint8_t *occ,
                                                        const int *flips, double *out_all) {
    const int lane = threadIdx.x;
    const int *fl = flips + (size_t)blockIdx.x * SMOLMC_STEP_ROW;
    double *out = out_all + (size_t)blockIdx.x * T.F;
    for (int i = lane; i < T.F; i += 64) out[i] = 0.0;
    __syncthreads();
    int nfl = 0;
    while (nfl < SMOLMC_MAX_STEP_FLIPS && fl[2 * nfl] >= 0) nfl++;
    auto seen = [&](const int x, const int f) -> int { // species of site x as flip f sees it
        int v = occ[x];
        for (int g = 0; g < f; ++g)
            if (fl[2 * g] == x) v = fl[2 * g + 1];
        return v;
    };
    double dew = 0, dmu = 0;
    for (int f = 0; f < nfl; ++f) {
        const int s = fl[2 * f], newc = fl[2 * f + 1];
        for (long long rr = T.site_ptr[s]; rr < T.site_ptr[s + 1]; ++rr) {
            const int n = T.loc_orbit[rr];
            const int I = T.orb_nsites[n], K = T.corr_mode ? T.orb_nfunc[n] : 1, Nt = T.orb_tensor_len[n];
            const int *st = T.tensor_indices + T.orb_stride_off[n];
            const int *ind = T.loc_idx + T.loc_off[rr];
            const int J = T.loc_nrows[rr];
            for (int k = 0; k < K; ++k) {
                const double *t = T.corr_mode ? T.corr_tensors + T.orb_ctensor_off[n] + (size_t)k * Nt
                                              : T.interaction_tensors + T.orb_itensor_off[n];
                double p = 0;
                for (int j = lane; j < J; j += 64) {
                    int ind_i = 0, ind_f = 0;
                    for (int i = 0; i < I; ++i) {
                        const int x = ind[j * I + i];
                        const int v = seen(x, f);
                        const int vf = (x == s) ? newc : v;
                        ind_i += st[i] * v;
                        ind_f += st[i] * vf;
                    }
                    p += t[ind_f] - t[ind_i];
                }
                p = wave_sum(p);
                if (lane == 0) {
                    const int o = T.corr_mode ? T.orb_bit_id[n] + k : T.orb_id[n];
                    out[o] += p / T.loc_ratio[rr] / (double)J;
                }
            }
        }
        if (T.has_ewald) {
            // ewald.pyx:38-58
            const int oldc = seen(s, f);
            const int W = T.ew_W;
            const int add = T.ew_inds[(size_t)s * W + newc], sub = T.ew_inds[(size_t)s * W + oldc];
            double o = 0;
            for (int k = lane; k < T.N; k += 64) {
                const int v = seen(k, f);
                const int vf = (k == s) ? newc : v;
                const int i = T.ew_inds[(size_t)k * W + vf], j = T.ew_inds[(size_t)k * W + v];
                if (i != -1 && add != -1)
                    o += (i != add ? 2.0 : 1.0) * T.ew_Mt[(size_t)add * T.ew_M + i];
                if (j != -1 && sub != -1)
                    o -= (j != sub ? 2.0 : 1.0) * T.ew_Mt[(size_t)sub * T.ew_M + j];
            }
            dew += wave_sum(o);
        }
        if (T.has_mu) // against the ORIGINAL occupancy (ensemble.py:368-374)
            dmu += T.mu[(size_t)s * T.mu_W + newc] - T.mu[(size_t)s * T.mu_W + occ[s]];
    }
    __syncthreads();
    for (int i = lane; i < T.Fce; i += 64) out[i] *= (double)T.P;
    if (lane == 0) {
        int f = T.Fce;
        if (T.has_ewald) out[f++] = dew;
        if (T.has_mu) out[f++] = dmu;
    }
}

__global__ void pack_occ_kernel(const int *occ32, uint8_t *occ8, int N, int Npad, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t rr = i / Npad;
    const int s = (int)(i % Npad);
    occ8[i] = s < N ? (uint8_t)occ32[rr * N + s] : 0;
}
__global__ void unpack_occ_kernel(const uint8_t *occ8, int *occ32, int N, int Npad, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t rr = i / N;
    const int s = (int)(i % N);
    occ32[i] = (int)occ8[rr * Npad + s];
}
// Ewald potential field of every walker from scratch (set_state): one block per walker, the
// charges q(k, occ_k) of the changeable sites staged in LDS, each wave sums rows
// phi[j] = sum_{k != j} q_k G[site_j][k].
__global__ void __launch_bounds__(256) ewald_field_init_kernel(const LeanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    double *q = (double *)fsm;
    const int r = blockIdx.x, na = P.ew_nact, W = P.ew_W;
    const uint8_t *occ = P.occ + (size_t)r * P.Npad;
    for (int k = threadIdx.x; k < na; k += blockDim.x) {
        const int site = P.sbase + k;
        q[k] = P.ew_qs[(size_t)site * W + occ[site]];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wave; j < na; j += nw) {
        const double *g = P.ew_G + (size_t)(P.sbase + j) * na;
        double acc = 0.0;
        for (int k = lane; k < na; k += 64) acc = fma(k != j ? q[k] : 0.0, g[k], acc);
        acc = wave_sum(acc);
        if (lane == 0) P.ew_phi[(size_t)r * na + j] = acc + P.ew_frozen[P.sbase + j];
    }
}

// Wang-Landau per-bin feature statistics: running means <-> running sums (sum = mean * occurrences)
__global__ void wl_meanf_convert_kernel(double *meanf, const long long *occur, size_t cells, int F, int to_mean) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells * (size_t)F) return;
    const long long n = occur[i / F];
    if (n > 0) meanf[i] = to_mean ? meanf[i] / (double)n : meanf[i] * (double)n;
}

__global__ void dot_features_kernel(const double *features, const double *natural, double *enthalpy,
                                    int R, int F) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double s = 0;
    for (int i = 0; i < F; ++i) s += natural[i] * features[(size_t)r * F + i];
    enthalpy[r] = s;
}

// ----------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------
static int num_ce_features(const smolmc_tables *t) {
    return t->feature_mode == SMOLMC_FEATURES_CORRELATIONS ? t->num_corr : t->num_orbits;
}

// Build the MC-optimised tables (classes, slot descriptors, member index rows).
// Models mc_kernel / the lean kernels cannot take (more than 1024 clusters per site, more than 255 site
// classes, tensor strides beyond 16 bits, an occupancy that does not fit LDS) are not refused: the
// handle notes why (general_reason) and every launch takes the universal kernel (mc_univ.h).
static int no_general(smolmc_handle *h, const char *why) {
    h->general_ok = false;
    h->general_reason = why;
    return 0;
}

static int build_mc_tables(smolmc_handle *h, const smolmc_tables *t) {
    const int N = t->num_sites;
    const bool corr = t->feature_mode == SMOLMC_FEATURES_CORRELATIONS;
    if ((size_t)h->Npad + 4096 > 160 * 1024) return no_general(h, "occupancy does not fit LDS");
    if (getenv("SMOLMC_FORCE_UNIVERSAL")) return no_general(h, "SMOLMC_FORCE_UNIVERSAL");
    // does any local row contain a repeated site (aliased tiny supercells)?
    bool aliased = false;
    int maxI = 1;
    for (int s = 0; s < N && !aliased; ++s)
        for (int64_t r = t->site_ptr[s]; r < t->site_ptr[s + 1] && !aliased; ++r) {
            const int o = t->loc_orbit[r], I = t->orb_nsites[o];
            const int32_t *rows = t->loc_idx + t->loc_off[r];
            for (int j = 0; j < t->loc_nrows[r] && !aliased; ++j)
                for (int a = 0; a < I && !aliased; ++a)
                    for (int b = a + 1; b < I; ++b)
                        if (rows[j * I + a] == rows[j * I + b]) aliased = true;
        }
    for (int o = 0; o < t->n_orb; ++o) maxI = std::max(maxI, (int)t->orb_nsites[o]);
    if (maxI > SMOLMC_MAX_CLUSTER_SITES) return fail("cluster larger than SMOLMC_MAX_CLUSTER_SITES");
    for (int o = 0; o < t->n_orb; ++o)
        for (int i = 0; i < t->orb_nsites[o]; ++i)
            if (t->tensor_indices[t->orb_stride_off[o] + i] > 65535)
                return no_general(h, "tensor stride exceeds 16 bits");
    h->generic = aliased;
    const int need_mm = aliased ? maxI : std::max(1, maxI - 1);

    // per-site slot lists: (orbit, self position p, row pointer, record r)
    struct Slot {
        int orbit, p, nmem;
        int64_t rec;
        const int32_t *row;
    };
    std::vector<std::vector<Slot>> slots(N);
    std::vector<int> site_class(N, 255);
    std::map<std::vector<long long>, int> class_of;
    std::vector<int> class_rep; // representative site per class
    for (int s = 0; s < N; ++s) {
        if (t->site_ptr[s] == t->site_ptr[s + 1]) continue;
        std::vector<Slot> &sl = slots[s];
        std::vector<long long> sig;
        for (int64_t r = t->site_ptr[s]; r < t->site_ptr[s + 1]; ++r) {
            const int o = t->loc_orbit[r], I = t->orb_nsites[o], J = t->loc_nrows[r];
            const int32_t *rows = t->loc_idx + t->loc_off[r];
            std::vector<Slot> rec;
            for (int j = 0; j < J; ++j) {
                int p = 0;
                if (!aliased)
                    for (int a = 0; a < I; ++a)
                        if (rows[j * I + a] == s) p = a;
                rec.push_back(Slot{o, p, aliased ? I : I - 1, r, rows + (size_t)j * I});
            }
            std::stable_sort(rec.begin(), rec.end(), [](const Slot &a, const Slot &b) { return a.p < b.p; });
            sig.push_back(o);
            sig.push_back(J);
            long long rb;
            memcpy(&rb, &t->loc_ratio[r], 8);
            sig.push_back(rb);
            for (int a = 0; a < I; ++a) {
                long long cnt = 0;
                for (auto &q : rec) cnt += q.p == a;
                sig.push_back(cnt);
            }
            sl.insert(sl.end(), rec.begin(), rec.end());
        }
        // most members first so that iterations are homogeneous
        std::stable_sort(sl.begin(), sl.end(), [](const Slot &a, const Slot &b) { return a.nmem > b.nmem; });
        auto itc = class_of.find(sig);
        if (itc == class_of.end()) {
            if (class_rep.size() >= 255) return no_general(h, "more than 255 site classes");
            itc = class_of.emplace(sig, (int)class_rep.size()).first;
            class_rep.push_back(s);
        }
        site_class[s] = itc->second;
    }
    const int nclasses = std::max<int>(1, (int)class_rep.size());
    size_t Cmax = 1;
    for (int s : class_rep) Cmax = std::max(Cmax, slots[s].size());
    const int niter_max = (int)((Cmax + 63) / 64);
    h->nslot = niter_max <= 2 ? 2 : (niter_max <= 4 ? 4 : (niter_max <= 8 ? 8 : 16));
    if (niter_max > 16) return no_general(h, "more than 1024 clusters per site");
    // slot columns per class: the two-group kernels evaluate BOTH groups without a condition
    // (mc_general.h), so the padding goes up to their full width (padded slots add 0.0)
    const int Cpad = 64 * (h->nslot <= 2 ? h->nslot : niter_max);
    if (aliased)
        h->mm = need_mm <= 3 ? 3 : 6;
    else
        h->mm = need_mm <= 2 ? 2 : (need_mm <= 3 ? 3 : 5);
    const int MM = h->mm;
    h->idx16 = (!aliased) && N <= 65535;

    // decision tensors per class, feature tensors shared
    std::vector<double> ft;
    std::vector<int> foff(t->n_orb);
    for (int o = 0; o < t->n_orb; ++o) {
        foff[o] = (int)ft.size();
        const int Nt = t->orb_tensor_len[o];
        if (corr) {
            const double *ct = t->corr_tensors + t->orb_ctensor_off[o];
            ft.insert(ft.end(), ct, ct + (size_t)t->orb_nfunc[o] * Nt);
        } else {
            const double *it = t->interaction_tensors + t->orb_itensor_off[o];
            ft.insert(ft.end(), it, it + Nt);
        }
    }
    std::vector<double> xt;
    std::vector<uint4> descA((size_t)nclasses * Cpad, make_uint4(0, 0, 0, 0));
    std::vector<uint4> descB((size_t)nclasses * Cpad, make_uint4(0, 0, 0, 0));
    std::vector<double> slot_fs((size_t)nclasses * Cpad, 0.0);
    std::vector<int> cls_niter(nclasses, 0);
    xt.push_back(0.0); // padded slots read xt[0] - xt[0]
    for (int c = 0; c < (int)class_rep.size(); ++c) {
        const int s = class_rep[c];
        const std::vector<Slot> &sl = slots[s];
        cls_niter[c] = (int)((sl.size() + 63) / 64);
        std::map<int64_t, int> xoff_of_rec;
        for (size_t q = 0; q < sl.size(); ++q) {
            const Slot &k = sl[q];
            const int o = k.orbit, I = t->orb_nsites[o], Nt = t->orb_tensor_len[o];
            const int32_t *st = t->tensor_indices + t->orb_stride_off[o];
            const double scale = (double)t->size / t->loc_ratio[k.rec] / (double)t->loc_nrows[k.rec];
            if (!xoff_of_rec.count(k.rec)) {
                xoff_of_rec[k.rec] = (int)xt.size();
                for (int i = 0; i < Nt; ++i) {
                    double v;
                    if (corr) {
                        // energy tensor: sum_k coef[bit_id+k] * ct[k][i], times scale
                        v = 0;
                        const double *ct = t->corr_tensors + t->orb_ctensor_off[o];
                        for (int kk = 0; kk < t->orb_nfunc[o]; ++kk)
                            v += t->ce_coefs[t->orb_bit_id[o] + kk] * ct[(size_t)kk * Nt + i];
                        v *= scale;
                    } else {
                        v = t->ce_coefs[t->orb_id[o]] * scale *
                            t->interaction_tensors[t->orb_itensor_off[o] + i];
                    }
                    xt.push_back(v);
                }
            }
            uint16_t sv[6] = {0, 0, 0, 0, 0, 0};
            if (aliased) {
                for (int a = 0; a < I; ++a) sv[a] = (uint16_t)st[a];
            } else {
                sv[0] = (uint16_t)st[k.p];
                int m = 1;
                for (int a = 0; a < I; ++a)
                    if (a != k.p) sv[m++] = (uint16_t)st[a];
            }
            uint4 A;
            A.x = (uint32_t)xoff_of_rec[k.rec];
            A.y = sv[0] | ((uint32_t)sv[1] << 16);
            A.z = sv[2] | ((uint32_t)sv[3] << 16);
            A.w = sv[4] | ((uint32_t)sv[5] << 16);
            descA[(size_t)c * Cpad + q] = A;
            uint4 B;
            B.x = (uint32_t)foff[o];
            B.y = (uint32_t)Nt;
            const uint32_t feat = corr ? (uint32_t)t->orb_bit_id[o] : (uint32_t)t->orb_id[o];
            const uint32_t K = corr ? (uint32_t)t->orb_nfunc[o] : 1u;
            B.z = feat | (K << 16);
            B.w = 0;
            descB[(size_t)c * Cpad + q] = B;
            slot_fs[(size_t)c * Cpad + q] = scale;
        }
    }
    if (xt.size() > 0xffffffffull) return no_general(h, "decision tensors too large");

    // member index rows [site][m][Cpad]; padded entries point at the site itself
    const size_t idx_n = (size_t)N * MM * Cpad;
    std::vector<int32_t> idx32(idx_n);
    for (int s = 0; s < N; ++s) {
        for (int m = 0; m < MM; ++m)
            for (int c = 0; c < Cpad; ++c) idx32[((size_t)s * MM + m) * Cpad + c] = s;
        const std::vector<Slot> &sl = slots[s];
        for (size_t q = 0; q < sl.size(); ++q) {
            const Slot &k = sl[q];
            const int I = t->orb_nsites[k.orbit];
            int m = 0;
            for (int a = 0; a < I; ++a) {
                if (!aliased && a == k.p) continue;
                idx32[((size_t)s * MM + m) * Cpad + q] = k.row[a];
                m++;
            }
        }
    }
    KParams &kp = h->kp;
    if (h->idx16) {
        std::vector<uint16_t> idx16(idx_n);
        for (size_t i = 0; i < idx_n; ++i) idx16[i] = (uint16_t)idx32[i];
        const uint16_t *d;
        TRY(dev_upload(h, idx16.data(), idx_n, &d));
        kp.idx = d;
    } else {
        const int32_t *d;
        TRY(dev_upload(h, idx32.data(), idx_n, &d));
        kp.idx = d;
    }
    std::vector<uint8_t> sc8(N);
    for (int s = 0; s < N; ++s) sc8[s] = (uint8_t)site_class[s];
    TRY(dev_upload(h, sc8.data(), (size_t)N, &kp.site_class));
    TRY(dev_upload(h, descA.data(), descA.size(), &kp.descA));
    TRY(dev_upload(h, descB.data(), descB.size(), &kp.descB));
    TRY(dev_upload(h, slot_fs.data(), slot_fs.size(), &kp.slot_fs));
    TRY(dev_upload(h, cls_niter.data(), cls_niter.size(), &kp.cls_niter));
    TRY(dev_upload(h, xt.data(), xt.size(), &kp.xt));
    TRY(dev_upload(h, ft.data(), ft.size(), &kp.ft));
    kp.xt_len = (int)xt.size();
    kp.ft_len = (int)ft.size();
    kp.nclasses = nclasses;
    kp.Cpad = Cpad;
    kp.Mmax = MM;

    // ---- lean tables (see mc_lean_kernel) ------------------------------------------
    memset(&h->lp, 0, sizeof(LeanParams));
    // The lean families' view of a site's clusters: one slot per local row, the flipped site's OWN positions in the row
    // folded into the slot's table (selfmask), the other members gathered.  On an unaliased cell a row holds the site
    // once (selfmask = 1 << p): exactly the slots above.  On an ALIASED cell -- a supercell shorter than a cluster, so
    // that a row holds a site twice; the reference keeps such rows (clusterspace.py:1353-1359) and its evaluator flips
    // every position of the site at once (evaluator.pyx:258-259 read occu_f / occu_i through the whole row) -- the
    // delta table of the slot does the same: D[(old, new)][b] = T[base(b) + (sum of the self strides) new] - T[... old]
    // (round 6; until then such cells ran on mc_kernel's GENERIC rows).  Order inside a record: by first self position,
    // then by mask -- the order of the reference's rows (equivalent cluster major) at equal positions, which is the same
    // on every site of a translation class; the classes are verified slot by slot below.
    struct LSlot {
        int orbit, pfirst, nother;
        uint32_t selfmask;
        int64_t rec;
        const int32_t *row;
    };
    std::vector<std::vector<LSlot>> lslots(N);
    int lean_need_mm = 1;
    for (int s = 0; s < N; ++s) {
        std::vector<LSlot> &sl = lslots[s];
        for (int64_t r = t->site_ptr[s]; r < t->site_ptr[s + 1]; ++r) {
            const int o = t->loc_orbit[r], I = t->orb_nsites[o], J = t->loc_nrows[r];
            const int32_t *rows = t->loc_idx + t->loc_off[r];
            std::vector<LSlot> rec;
            for (int j = 0; j < J; ++j) {
                uint32_t mask = 0;
                for (int a = 0; a < I; ++a)
                    if (rows[j * I + a] == s) mask |= 1u << a;
                const int pf = mask ? __builtin_ctz(mask) : 0;
                rec.push_back(LSlot{o, pf, I - __builtin_popcount(mask), mask, r, rows + (size_t)j * I});
            }
            std::stable_sort(rec.begin(), rec.end(), [](const LSlot &a, const LSlot &b) {
                return a.pfirst != b.pfirst ? a.pfirst < b.pfirst : a.selfmask < b.selfmask;
            });
            sl.insert(sl.end(), rec.begin(), rec.end());
        }
        std::stable_sort(sl.begin(), sl.end(), [](const LSlot &a, const LSlot &b) { return a.nother > b.nother; });
        for (const LSlot &q : sl) lean_need_mm = std::max(lean_need_mm, q.nother);
    }
    // aliased cells: every site of a class must list the same slots as the class's representative
    bool lean_slots_ok = true;
    if (aliased)
        for (int s = 0; s < N && lean_slots_ok; ++s) {
            if (site_class[s] == 255) continue;
            const std::vector<LSlot> &a = lslots[s], &b = lslots[class_rep[site_class[s]]];
            lean_slots_ok = a.size() == b.size();
            for (size_t q = 0; q < a.size() && lean_slots_ok; ++q)
                lean_slots_ok = a[q].orbit == b[q].orbit && a[q].selfmask == b[q].selfmask && a[q].selfmask != 0;
        }
    const bool lean_aliased_ok = !aliased || (lean_slots_ok && getenv("SMOLMC_NO_LEAN_ALIASED") == nullptr);
    // one site class: mc_lean_kernel (NSLOT <= 4); up to four classes or up to 512 clusters per
    // site: mc_lean_multi_kernel (per-class slot records in LDS)
    // Correlation features (ClusterExpansionProcessor, evaluator.pyx:211-265): when every orbit
    // has a single correlation function (K = 1: every binary system) the correlation delta of a
    // cluster is one tensor difference exactly like the interaction delta (:267-317), so the lean
    // kernels serve it with delta tables made from the correlation tensor, the feature index
    // bit_id instead of the orbit id and the weight coefs[bit_id]: same code, other tables.
    bool corr_k1 = corr;
    for (int o = 0; o < t->n_orb; ++o) corr_k1 = corr_k1 && t->orb_nfunc[o] == 1;
    // Several correlation functions per orbit (K <= SMOLMC_LEAN_MAX_KF, one site class): the decision
    // table of a slot is made from the folded tensor E = sum_k coef_k ct_k and the K function
    // tables follow it (mc_lean_kernel KF).
    int kmax = 1;
    for (int o = 0; o < t->n_orb; ++o) kmax = std::max(kmax, (int)t->orb_nfunc[o]);
    const bool cfg_wl = h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU;
    // (the KF instantiations are plain Metropolis flip / swap kernels of the single-class layout)
    bool corr_kf = corr && !corr_k1 && kmax <= SMOLMC_LEAN_MAX_KF && class_rep.size() == 1 && !cfg_wl && !t->bias_type &&
                   h->cfg.step_type != SMOLMC_STEP_TABLE_FLIP && t->n_sublattices == 1 && niter_max <= 4 && num_ce_features(t) <= 64 &&
                   getenv("SMOLMC_LAZY_FEATURES_ONLY") == nullptr;
    // ... and the Wang-Landau kernel of the multi-class layout (KFW, round 5): any number of classes the layout takes
    if (corr && !corr_k1 && kmax <= SMOLMC_LEAN_MAX_KF && cfg_wl && !t->bias_type && h->cfg.step_type != SMOLMC_STEP_TABLE_FLIP &&
        num_ce_features(t) <= 61 && getenv("SMOLMC_NO_WL_KF") == nullptr)
        corr_kf = true;
    // LAZY cluster features (round 5): every other Metropolis kernel of the lean families takes a model with several
    // correlation functions per orbit as an interaction-mode model of the folded tensors E = sum_k coef_k ct_k --
    // the decision needs nothing else -- and carries no cluster features at all: they are evaluated from the
    // occupancy where somebody reads them (smolmc_get_state, sample rows; ensure_features / lazy_rows_kernel).
    // Any number of functions per orbit and of cluster features; Wang-Landau needs the features on every step and
    // stays on mc_kernel.
    bool corr_lazy = corr && !corr_k1 && !corr_kf && !cfg_wl && getenv("SMOLMC_NO_LAZY_FEATURES") == nullptr;
    if (getenv("SMOLMC_NO_LEAN_CORR")) corr_k1 = corr_kf = corr_lazy = false; // A/B switch (tests, profiling)
    // ... and the same for models of more than 64 cluster features (the kernels' feature cells are one per lane)
    const bool wide_lazy = !corr_lazy && !corr_kf && (!corr || corr_k1) && num_ce_features(t) > 64 && !cfg_wl &&
                           getenv("SMOLMC_NO_LAZY_FEATURES") == nullptr;
    const bool lazy_any = corr_lazy || wide_lazy;
    // (why a model does not get the lean tables, reported by smolmc_kernel_info: the first condition that fails)
    h->lean_reason = class_rep.size() < 1 ? "no site with clusters"
                     : class_rep.size() > 4 ? "more than 4 site classes"
                     : !lean_aliased_ok ? "aliased supercell (a cluster holds a site twice) whose sites do not list their clusters alike"
                     : (corr && !corr_k1 && !corr_kf && !corr_lazy) ? (cfg_wl ? "Wang-Landau with more than SMOLMC_LEAN_MAX_KF correlation functions per orbit, more than 61 of them, or TableFlip"
                                                                                : "environment override (SMOLMC_NO_LEAN_CORR / SMOLMC_NO_LAZY_FEATURES)")
                     : N > 65535 ? "more than 65535 sites"
                     : niter_max > 8 ? "more than 512 clusters per site"
                     : lean_need_mm > 3 ? "clusters of more than 4 sites"
                     : (num_ce_features(t) > 64 && !lazy_any) ? "more than 64 cluster features (Wang-Landau or correlation functions on the KF kernels)" : "";
    if (class_rep.size() >= 1 && class_rep.size() <= 4 && lean_aliased_ok && (!corr || corr_k1 || corr_kf || corr_lazy) && N <= 65535 &&
        niter_max <= 8 && lean_need_mm <= 3 && (num_ce_features(t) <= 64 || lazy_any)) {
        const int NSL = niter_max <= 2 ? 2 : (niter_max <= 4 ? 4 : 8);
        const int NCLS = (int)class_rep.size();
        const int MML = lean_need_mm <= 2 ? 2 : 3;
        const int ROW = NSL * MML;
        std::vector<uint16_t> lidx((size_t)N * 64 * ROW);
        for (int s = 0; s < N; ++s) {
            for (int q = 0; q < 64 * ROW; ++q) lidx[(size_t)s * 64 * ROW + q] = (uint16_t)s;
            const std::vector<LSlot> &sl = lslots[s];
            for (size_t q = 0; q < sl.size(); ++q) {
                const LSlot &k = sl[q];
                const int I = t->orb_nsites[k.orbit];
                const int it = (int)(q / 64), ln = (int)(q % 64);
                int m = 0;
                for (int a = 0; a < I; ++a) {
                    if ((k.selfmask >> a) & 1u) continue; // (the site's own positions are folded into the slot's table)
                    lidx[(((size_t)s * 64 + ln) * NSL + it) * MML + m] = (uint16_t)k.row[a];
                    m++;
                }
            }
        }
        // ---- LDS bank swizzle: pick the address permutation s ^ (((s >> a) & m) << b) that
        // minimises the modelled bank-conflict cycles of the occupancy gathers (each
        // ds_read_u8 is served in two 32-lane groups; a group costs the max number of
        // distinct dwords on one of the 32 banks).  See tools/lds_conflict_model.py.
        int Nlds = 16;
        while (Nlds < h->Npad) Nlds <<= 1;
        int best_a = 0, best_m = 0, best_b = 0;
        {
            auto model_cost = [&](int a, int m, int b) {
                double tot = 0;
                const int nsamp = std::min(N, 48);
                for (int k = 0; k < nsamp; ++k) {
                    const int s = (int)(((long long)k * 2654435761ll) % N);
                    if (lslots[s].empty()) continue;
                    for (int q = 0; q < ROW; ++q)
                        for (int g = 0; g < 2; ++g) {
                            int cnt[32] = {0};
                            int seen[32];
                            int nseen = 0;
                            for (int ln = 32 * g; ln < 32 * g + 32; ++ln) {
                                const int x = lidx[((size_t)s * 64 + ln) * ROW + q];
                                const int dw = (x ^ (((x >> a) & m) << b)) >> 2;
                                bool dup = false;
                                for (int z = 0; z < nseen; ++z) dup |= seen[z] == dw;
                                if (!dup) { seen[nseen++] = dw; cnt[dw & 31]++; }
                            }
                            int mx = 0;
                            for (int z = 0; z < 32; ++z) mx = std::max(mx, cnt[z]);
                            tot += mx;
                        }
                }
                return tot;
            };
            double best = model_cost(0, 0, 0);
            const int amax = getenv("SMOLMC_NO_SWIZZLE") ? 0 : 12; // A/B switch for profiling
            for (int a = 3; a <= amax; ++a)
                for (int b = 2; b <= 5; ++b)
                    for (int m : {3, 7, 15, 31}) {
                        // bijection on [0, Nlds): source bits [a, a+k) above the destination
                        // bits [b, b+k) and inside the (power-of-two) array
                        const int k = m == 3 ? 2 : (m == 7 ? 3 : (m == 15 ? 4 : 5));
                        if (a < b + k || (1 << (a + k)) > Nlds) continue;
                        const double c = model_cost(a, m, b);
                        if (c < best * 0.97) { best = c; best_a = a; best_m = m; best_b = b; }
                    }
        }
        if (best_m == 0) Nlds = h->Npad; // identity: no power-of-two padding needed
        for (size_t i = 0; i < lidx.size(); ++i) {
            const int x = lidx[i];
            lidx[i] = (uint16_t)(x ^ (((x >> best_a) & best_m) << best_b));
        }
        h->lp.swz_a = best_a; h->lp.swz_m = best_m; h->lp.swz_b = best_b; h->lp.Nlds = Nlds;

        // delta tables, one per (orbit, self position): D[(old, new)][b] with the COMPACT base
        // index b = sum_m S^m * species(member m) over the other members of the cluster (S =
        // max species per site), padded to a common [S*S][NTP], NTP = S^MML.  Symmetric
        // clusters give bitwise-equal tables for several self positions: those are shared.
        const int SMAX = t->max_species;
        int NTP = 1;
        for (int m = 0; m < MML; ++m) NTP *= SMAX;
        // the stride between tables is padded so that it is not a multiple of the 64-dword LDS
        // bank period (lanes of one wave read the same (pair, base) entry of DIFFERENT tables)
        size_t tlen = (size_t)SMAX * SMAX * NTP;
        if ((tlen & 1) == 0) tlen += 1;
        std::vector<double> dt(tlen, 0.0); // table 0 = zeros, used by padded slots
        // Entries the LDS copy of the tables may take.  8000 (64 KB, two workgroups per CU) until round 6; models between
        // that and what one workgroup's LDS holds beside the walkers' state (the layouts below decide) ran on mc_kernel
        // with the tables in L2: measured on five quaternary triplet / quadruplet models of the fuzz campaign at 2048
        // walkers, 0.39-1.05e9 steps/s there against 0.99-2.43e9 here (tools/time_fuzz_case.py).  SMOLMC_LEAN_DT_MAX: A/B hook.
        const size_t dt_max = getenv("SMOLMC_LEAN_DT_MAX") ? (size_t)atol(getenv("SMOLMC_LEAN_DT_MAX")) : (size_t)18000;
        std::vector<double> dtk; // KF: correlation-function tables (global memory), see LeanParams::dtk
        std::vector<LeanSlot> ls((size_t)NCLS * NSL * 64);
        memset(ls.data(), 0, ls.size() * sizeof(LeanSlot));
        std::map<std::pair<int, int>, uint32_t> doff_of;
        bool ok = true;
        double sum_abs_max = 0.0;
        for (int cls = 0; cls < NCLS && ok; ++cls) {
        const std::vector<LSlot> &sl = lslots[class_rep[cls]];
        double sum_abs = 0.0;
        for (size_t q = 0; q < sl.size() && ok; ++q) {
            const LSlot &k = sl[q];
            const int o = k.orbit, I = t->orb_nsites[o], Nt = t->orb_tensor_len[o];
            const int32_t *st = t->tensor_indices + t->orb_stride_off[o];
            const double *T = corr ? t->corr_tensors + t->orb_ctensor_off[o] // K == 1: the one function
                                   : t->interaction_tensors + t->orb_itensor_off[o];
            const int feat = corr ? t->orb_bit_id[o] : t->orb_id[o];
            const int K = corr_kf ? t->orb_nfunc[o] : 1;
            const int Kfold = (corr_kf || corr_lazy) ? t->orb_nfunc[o] : 1;
            std::vector<double> Efold; // KF / lazy: folded tensor sum_k coef_k ct_k (decision table source)
            if (corr_kf || corr_lazy) {
                Efold.assign((size_t)Nt, 0.0);
                for (int kk = 0; kk < Kfold; ++kk)
                    for (int i = 0; i < Nt; ++i) Efold[i] += t->ce_coefs[feat + kk] * T[(size_t)kk * Nt + i];
            }
            // the site's own positions: their strides add up (one position on an unaliased cell), its site space is
            // that of the first of them
            int ss = 0;
            for (int a = 0; a < I; ++a)
                if ((k.selfmask >> a) & 1u) ss += st[a];
            const int Sself = k.pfirst == 0 ? Nt / st[0] : st[k.pfirst - 1] / st[k.pfirst];
            const int nother = k.nother;
            const auto key = std::make_pair(o, (int)k.selfmask);
            if (!doff_of.count(key)) {
                // tables of the slot: the decision table (LDS) and, in KF mode, K correlation-function
                // tables (global memory, read on accepted steps only)
                const int ntab = corr_kf ? 1 + K : 1;
                std::vector<double> D((size_t)ntab * tlen, 0.0);
                int nb = 1;
                for (int a = 0; a < nother; ++a) nb *= SMAX;
                for (int tb = 0; tb < ntab; ++tb) {
                const double *Tsrc = corr_lazy ? Efold.data() : !corr_kf ? T : (tb == 0 ? Efold.data() : T + (size_t)(tb - 1) * Nt);
                double *Dt = D.data() + (size_t)tb * tlen;
                for (int b = 0; b < nb; ++b) {
                    // decode b into the species of the other members -> tensor base index
                    long base = 0;
                    int rem = b;
                    bool valid = true;
                    for (int a = 0; a < I; ++a) {
                        if ((k.selfmask >> a) & 1u) continue;
                        const int v = rem % SMAX;
                        rem /= SMAX;
                        const int Sa = a == 0 ? Nt / st[0] : st[a - 1] / st[a];
                        if (v >= Sa) valid = false;
                        base += (long)st[a] * v;
                    }
                    if (!valid) continue;
                    for (int oldc = 0; oldc < Sself; ++oldc)
                        for (int newc = 0; newc < Sself; ++newc)
                            Dt[((size_t)oldc * SMAX + newc) * NTP + b] =
                                Tsrc[base + (long)ss * newc] - Tsrc[base + (long)ss * oldc];
                }
                }
                uint32_t at = 0; // (identical tables are shared; KF: decision AND function tables identical)
                for (size_t off = tlen; off + tlen <= dt.size() && !at; off += tlen) {
                    if (memcmp(dt.data() + off, D.data(), (size_t)tlen * sizeof(double)) != 0) continue;
                    if (corr_kf) {
                        std::vector<double> want((size_t)SMOLMC_LEAN_MAX_KF * tlen, 0.0);
                        std::copy(D.begin() + tlen, D.end(), want.begin());
                        if (memcmp(dtk.data() + off * SMOLMC_LEAN_MAX_KF, want.data(), want.size() * sizeof(double)) != 0)
                            continue;
                    }
                    at = (uint32_t)off;
                }
                if (!at) {
                    at = (uint32_t)dt.size();
                    dt.insert(dt.end(), D.begin(), D.begin() + tlen);
                    if (corr_kf) {
                        dtk.resize(dt.size() * SMOLMC_LEAN_MAX_KF, 0.0);
                        std::copy(D.begin() + tlen, D.end(), dtk.begin() + (size_t)at * SMOLMC_LEAN_MAX_KF);
                    }
                }
                doff_of[key] = at;
            }
            const double scale = (double)t->size / t->loc_ratio[k.rec] / (double)t->loc_nrows[k.rec];
            LeanSlot &L = ls[((size_t)cls * NSL + q / 64) * 64 + (q % 64)];
            L.doff8 = doff_of[key] * 8u;
            {
                uint32_t cs = 8u;
                for (int m = 0; m < nother; ++m, cs *= (uint32_t)SMAX) L.stride8[m] = cs;
            }
            L.feat = lazy_any ? 0u : (uint32_t)feat; // (lazy: the kernels' feature cells are never read)
            L.live = (uint32_t)K;
            L.w = (corr_kf || corr_lazy) ? scale : t->ce_coefs[feat] * scale; // (KF / lazy: the coefficients are folded into the table)
            L.fs = scale;
            if (dt.size() > dt_max) { ok = false; h->lean_reason = "delta tables beyond 18000 entries (species^(cluster size - 1) x species^2 per distinct table)"; } // keep the LDS tables within budget
            double dmax = 0.0;
            {
                const double *D = dt.data() + doff_of[key];
                for (size_t z = 0; z < tlen; ++z) dmax = std::max(dmax, std::fabs(D[z]));
            }
            sum_abs += std::fabs(L.w) * dmax;
        }
        sum_abs_max = std::max(sum_abs_max, sum_abs);
        }
        // float32 pre-test of the accept decision: a step sums at most two flips' worth of
        // |w * d| over the slots, a float32 conversion + 6-level tree adds at most
        // 7 * 2^-24 of that; 2^-19 leaves a 4.5x margin.
        h->lp.fast_eps = 2.0 * sum_abs_max * ldexp(1.0, -19);
        h->lp.ktab8 = (uint32_t)tlen * 8u;
        h->lean_kf = corr_kf ? SMOLMC_LEAN_MAX_KF : 0;
        h->lazy_tables = lazy_any;
        h->lp.nt8 = (uint32_t)NTP * 8u;
        h->lp.snt8 = (uint32_t)NTP * 8u * (uint32_t)SMAX;
        if (ok) {
            TRY(dev_upload(h, lidx.data(), lidx.size(), &h->lp.idx));
            h->lean_idx_host = std::move(lidx);
            TRY(dev_upload(h, dt.data(), dt.size(), &h->lp.dt));
            if (corr_kf) TRY(dev_upload(h, dtk.data(), dtk.size(), &h->lp.dtk));
            TRY(dev_upload(h, ls.data(), ls.size(), &h->lp.slots));
            h->lp.dt_len = (int)dt.size();
            h->lean_tables = true;
            h->lean_nslot = NSL;
            h->lean_mm = MML;
            h->lean_ncls = NCLS;
            h->site_class_host = site_class;
        }
    }
    return 0;
}

// Bounds checks of every gather index the kernels will use, done once at create (the
// reference's compiled core runs with boundscheck=False and reads garbage for a bad index).  The
// kernels themselves never check: every index they form comes from these tables and from occupancy
// codes, which smolmc_set_state checks against the tables.  -DSMOLMC_BOUNDS adds device-side traps
// at the gathers of the lean kernels for debugging the kernels' own address arithmetic.
static int validate_tables(const smolmc_tables *t) {
    const int N = t->num_sites, n = t->n_orb;
    if (N <= 0 || n < 0 || t->size <= 0) return fail("num_sites / size / n_orb must be positive");
    int nstr = 0;
    for (int o = 0; o < n; ++o) {
        if (t->orb_nsites[o] <= 0 || t->orb_nfunc[o] <= 0 || t->orb_tensor_len[o] <= 0)
            return fail("orbit record with a non-positive size");
        if (t->orb_stride_off[o] != nstr) return fail("orb_stride_off does not follow the orbit sizes");
        nstr += t->orb_nsites[o];
        // the largest flat tensor index a cluster of this orbit can form must lie inside the tensor
        // (strides of a row-major tensor over the site spaces of the cluster, orbit.py:268-275: member m
        // has orb_tensor_len / stride_0 species for m = 0 and stride_{m-1} / stride_m after that; the
        // largest flat index sum_m (species_m - 1) stride_m is then orb_tensor_len - 1)
        long long prev = t->orb_tensor_len[o], reach = 0;
        for (int m = 0; m < t->orb_nsites[o]; ++m) {
            const int st = t->tensor_indices[t->orb_stride_off[o] + m];
            if (st <= 0) return fail("tensor stride must be positive");
            if (st > prev || prev % st != 0) return fail("tensor strides do not describe a row-major tensor");
            reach += (prev / st - 1) * st;
            prev = st;
        }
        if (reach >= t->orb_tensor_len[o]) return fail("tensor strides reach beyond their tensor");
        const int64_t a = t->full_off[o], b = t->full_off[o + 1];
        if (b < a || (b - a) % t->orb_nsites[o] != 0) return fail("full_off does not describe whole cluster rows");
        for (int64_t i = a; i < b; ++i)
            if (t->full_idx[i] < 0 || t->full_idx[i] >= N) return fail("cluster site index out of range (full table)");
    }
    for (int s = 0; s < N; ++s)
        if (t->site_ptr[s + 1] < t->site_ptr[s]) return fail("site_ptr must be non-decreasing");
    const int64_t nloc = t->site_ptr[N];
    for (int64_t r = 0; r < nloc; ++r) {
        const int o = t->loc_orbit[r];
        if (o < 0 || o >= n) return fail("local record refers to an orbit out of range");
        if (t->loc_nrows[r] <= 0 || !(t->loc_ratio[r] > 0)) return fail("local record with a non-positive size / ratio");
        const int64_t cnt = (int64_t)t->loc_nrows[r] * t->orb_nsites[o];
        for (int64_t i = 0; i < cnt; ++i) {
            const int v = t->loc_idx[t->loc_off[r] + i];
            if (v < 0 || v >= N) return fail("cluster site index out of range (local table)");
        }
    }
    if (t->has_ewald) {
        if (t->ewald_width <= 0 || t->ewald_dim <= 0) return fail("Ewald table with a non-positive size");
        for (int64_t i = 0; i < (int64_t)N * t->ewald_width; ++i)
            if (t->ewald_inds[i] < -1 || t->ewald_inds[i] >= t->ewald_dim) return fail("Ewald index out of range");
    }
    if (t->has_mu && t->mu_width <= 0) return fail("chemical-potential table with a non-positive width");
    return 0;
}

static int build_ref_tables(smolmc_handle *h, const smolmc_tables *t) {
    RefTables &rt = h->rt;
    memset(&rt, 0, sizeof(rt));
    const int n = t->n_orb;
    rt.N = t->num_sites;
    rt.Npad = h->Npad;
    rt.P = t->size;
    rt.num_orbits = t->num_orbits;
    rt.num_corr = t->num_corr;
    rt.n_orb = n;
    rt.Fce = h->Fce;
    rt.F = h->F;
    rt.corr_mode = t->feature_mode == SMOLMC_FEATURES_CORRELATIONS;
    rt.offset = t->offset;
    int nstr = 0;
    long long nct = 0, nit = 0;
    for (int o = 0; o < n; ++o) {
        nstr += t->orb_nsites[o];
        nct += (long long)t->orb_nfunc[o] * t->orb_tensor_len[o];
        nit += t->orb_tensor_len[o];
    }
    const int64_t nloc = t->site_ptr[t->num_sites];
    int64_t nlocidx = 0;
    for (int64_t r = 0; r < nloc; ++r)
        nlocidx += (int64_t)t->loc_nrows[r] * t->orb_nsites[t->loc_orbit[r]];
    TRY(dev_upload(h, t->orb_id, n, &rt.orb_id));
    TRY(dev_upload(h, t->orb_bit_id, n, &rt.orb_bit_id));
    TRY(dev_upload(h, t->orb_nsites, n, &rt.orb_nsites));
    TRY(dev_upload(h, t->orb_nfunc, n, &rt.orb_nfunc));
    TRY(dev_upload(h, t->orb_tensor_len, n, &rt.orb_tensor_len));
    TRY(dev_upload(h, t->orb_stride_off, n, &rt.orb_stride_off));
    TRY(dev_upload(h, t->tensor_indices, nstr, &rt.tensor_indices));
    TRY(dev_upload(h, (const long long *)t->orb_ctensor_off, n, &rt.orb_ctensor_off));
    TRY(dev_upload(h, (const long long *)t->orb_itensor_off, n, &rt.orb_itensor_off));
    TRY(dev_upload(h, t->corr_tensors, nct, &rt.corr_tensors));
    TRY(dev_upload(h, t->interaction_tensors, nit, &rt.interaction_tensors));
    TRY(dev_upload(h, (const long long *)t->full_off, n + 1, &rt.full_off));
    TRY(dev_upload(h, t->full_idx, t->full_off[n], &rt.full_idx));
    TRY(dev_upload(h, (const long long *)t->site_ptr, t->num_sites + 1, &rt.site_ptr));
    TRY(dev_upload(h, t->loc_orbit, nloc, &rt.loc_orbit));
    TRY(dev_upload(h, t->loc_ratio, nloc, &rt.loc_ratio));
    TRY(dev_upload(h, t->loc_nrows, nloc, &rt.loc_nrows));
    TRY(dev_upload(h, (const long long *)t->loc_off, nloc, &rt.loc_off));
    TRY(dev_upload(h, t->loc_idx, nlocidx, &rt.loc_idx));
    rt.has_ewald = t->has_ewald;
    rt.has_mu = t->has_mu;
    if (t->has_ewald) {
        const size_t M = t->ewald_dim;
        rt.ew_W = t->ewald_width;
        rt.ew_M = (int)M;
        TRY(dev_upload(h, t->ewald_inds, (size_t)t->num_sites * t->ewald_width, &rt.ew_inds));
        TRY(dev_upload(h, t->ewald_matrix, M * M, &rt.ew_M_rowmajor));
        std::vector<double> mt(M * M);
        const size_t B = 64;
        for (size_t i0 = 0; i0 < M; i0 += B)
            for (size_t j0 = 0; j0 < M; j0 += B)
                for (size_t i = i0; i < std::min(M, i0 + B); ++i)
                    for (size_t j = j0; j < std::min(M, j0 + B); ++j)
                        mt[j * M + i] = t->ewald_matrix[i * M + j];
        TRY(dev_upload(h, mt.data(), M * M, &rt.ew_Mt));
    }
    if (t->has_mu) {
        rt.mu_W = t->mu_width;
        TRY(dev_upload(h, t->mu_table, (size_t)t->num_sites * t->mu_width, &rt.mu));
    }
    return 0;
}

// Check M[a][b] == q_a q_b G[site_a][site_b] (a, b on different sites) and build G.
// Translation-compressed site kernel.  On a supercell of a periodic lattice the site
// kernel G[s][j] of the compact Ewald form depends on the sublattices of s and j and on the lattice
// translation between them only (smol/utils/cluster/ewald.pyx:38-58 reads the same numbers N times
// over).  When the changeable sites come as contiguous blocks of P = `size` sites in the
// lexicographic order of the translation coordinates of a d1 x d2 x d3 supercell -- the order of
// pymatgen's lattice_points_in_supercell (and smol_amd.synth) for diagonal supercell matrices; no
// coordinates travel through the C ABI, the structure is found and then VERIFIED on every entry of
// G -- one table per block pair over the coordinate differences -(d-1) .. d-1 replaces the rows in the
// potential-field updates (field_sweep_gx, smolmc_common.h): entry (s, j) at byte E8[j] + S8[s].
// Anything else (non-diagonal supercells, other site orders) keeps the rows of G.
static int compress_ewald_rows(smolmc_handle *h, const smolmc_tables *t, const std::vector<double> &Gact,
                               const std::vector<int> &act) {
    const size_t na = act.size();
    const int P = t->size;
    if (getenv("SMOLMC_NO_EWALD_GX") != nullptr || P <= 1 || na == 0 || na % (size_t)P != 0) return 0;
    const int nbk = (int)(na / (size_t)P);
    for (size_t j = 0; j < na; ++j)
        if (act[j] != act[0] + (int)j) return 0; // (the field mode needs contiguous changeable sites anyway)
    auto Gat = [&](size_t js, size_t j) { return Gact[(size_t)act[js] * na + j]; };
    double gmax = 0.0;
    for (size_t js = 0; js < na; ++js)
        for (size_t j = 0; j < na; ++j) gmax = std::max(gmax, std::fabs(Gat(js, j)));
    const double tol = 1e-12 * std::max(gmax, 1e-300);
    // entry (block k, translation t) x (block k2, translation t2) against row "translation 0" of
    // block k at the wrapped coordinate difference
    auto rows_match = [&](const int d[3], int k, int tt) {
        const int x = tt / (d[1] * d[2]), y = (tt / d[2]) % d[1], z = tt % d[2];
        for (int k2 = 0; k2 < nbk; ++k2)
            for (int t2 = 0; t2 < P; ++t2) {
                const int x2 = t2 / (d[1] * d[2]), y2 = (t2 / d[2]) % d[1], z2 = t2 % d[2];
                const int rel = (((x2 - x + d[0]) % d[0]) * d[1] + (y2 - y + d[1]) % d[1]) * d[2] + (z2 - z + d[2]) % d[2];
                if (std::fabs(Gat((size_t)k * P + tt, (size_t)k2 * P + t2) - Gat((size_t)k * P, (size_t)k2 * P + rel)) > tol)
                    return false;
            }
        return true;
    };
    // factorisations d1 d2 d3 = P, the most balanced first; a few rows screen a candidate, then
    // every row confirms it
    std::vector<std::array<int, 3>> cand;
    for (int a = 1; a <= P; ++a)
        if (P % a == 0)
            for (int b = 1; b <= P / a; ++b)
                if ((P / a) % b == 0) cand.push_back({a, b, P / a / b});
    std::sort(cand.begin(), cand.end(), [](const std::array<int, 3> &u, const std::array<int, 3> &v) {
        auto spread = [](const std::array<int, 3> &w) { return std::max({w[0], w[1], w[2]}) - std::min({w[0], w[1], w[2]}); };
        return spread(u) < spread(v);
    });
    int d[3] = {0, 0, 0};
    bool found = false;
    for (const auto &c : cand) {
        const int dd[3] = {c[0], c[1], c[2]};
        bool ok = true;
        for (int probe : {1, P / 2 + 1, P - 1, (int)((2654435761u % (unsigned)P))})
            if (ok && probe > 0 && probe < P) ok = rows_match(dd, nbk - 1, probe);
        for (int k = 0; ok && k < nbk; ++k)
            for (int tt = 0; ok && tt < P; ++tt) ok = rows_match(dd, k, tt);
        if (ok) { d[0] = dd[0]; d[1] = dd[1]; d[2] = dd[2]; found = true; break; }
    }
    if (!found) return 0;
    const int e[3] = {2 * d[0] - 1, 2 * d[1] - 1, 2 * d[2] - 1};
    const size_t EXT = (size_t)e[0] * e[1] * e[2];
    if ((size_t)nbk * nbk * EXT * 8 >= ((size_t)1 << 31)) return 0; // 32-bit byte offsets
    std::vector<double> gx((size_t)nbk * nbk * EXT);
    for (int k = 0; k < nbk; ++k)
        for (int k2 = 0; k2 < nbk; ++k2)
            for (int a = 0; a < e[0]; ++a)
                for (int b = 0; b < e[1]; ++b)
                    for (int c = 0; c < e[2]; ++c) {
                        const int rel = (((a - (d[0] - 1) + d[0]) % d[0]) * d[1] + (b - (d[1] - 1) + d[1]) % d[1]) * d[2] +
                                        (c - (d[2] - 1) + d[2]) % d[2];
                        gx[((size_t)k * nbk + k2) * EXT + ((size_t)a * e[1] + b) * e[2] + c] = Gat((size_t)k * P, (size_t)k2 * P + rel);
                    }
    const size_t center = ((size_t)(d[0] - 1) * e[1] + (d[1] - 1)) * e[2] + (d[2] - 1);
    std::vector<uint32_t> E8(na), S8(na);
    for (int k = 0; k < nbk; ++k)
        for (int tt = 0; tt < P; ++tt) {
            const int x = tt / (d[1] * d[2]), y = (tt / d[2]) % d[1], z = tt % d[2];
            const size_t E = ((size_t)x * e[1] + y) * e[2] + z;
            E8[(size_t)k * P + tt] = (uint32_t)(8 * ((size_t)k * EXT + E));
            S8[(size_t)k * P + tt] = (uint32_t)(8 * ((size_t)k * nbk * EXT + center - E));
        }
    TRY(dev_upload(h, gx.data(), gx.size(), &h->kp.ew_gx));
    TRY(dev_upload(h, E8.data(), E8.size(), &h->kp.ew_E8));
    TRY(dev_upload(h, S8.data(), S8.size(), &h->kp.ew_S8));
    h->ew_gx_dims[0] = d[0]; h->ew_gx_dims[1] = d[1]; h->ew_gx_dims[2] = d[2];
    h->ew_gx_blocks = nbk;
    return 0;
}

static int build_compact_ewald(smolmc_handle *h, const smolmc_tables *t) {
    const int N = t->num_sites, W = t->ewald_width;
    const size_t M = (size_t)t->ewald_dim;
    const double *Mx = t->ewald_matrix, *q = t->ewald_charges;
    std::vector<double> G((size_t)N * N, 0.0), qs((size_t)N * W, 0.0), dg((size_t)N * W, 0.0);
    double mmax = 0;
    for (size_t i = 0; i < M * M; ++i) mmax = std::max(mmax, fabs(Mx[i]));
    const double tol = 1e-12 * std::max(mmax, 1e-300);
    for (int s = 0; s < N; ++s)
        for (int c = 0; c < W; ++c) {
            const int a = t->ewald_inds[(size_t)s * W + c];
            if (a < 0) continue;
            qs[(size_t)s * W + c] = q[a];
            dg[(size_t)s * W + c] = Mx[(size_t)a * M + a];
        }
    bool ok = true;
    for (int s = 0; s < N && ok; ++s)
        for (int u = 0; u < N && ok; ++u) {
            if (s == u) continue;
            double g = 0;
            bool have = false;
            for (int c = 0; c < W && !have; ++c)
                for (int d = 0; d < W && !have; ++d) {
                    const int a = t->ewald_inds[(size_t)s * W + c], b = t->ewald_inds[(size_t)u * W + d];
                    if (a < 0 || b < 0 || q[a] == 0.0 || q[b] == 0.0) continue;
                    g = Mx[(size_t)b * M + a] / (q[a] * q[b]); // the entry ewald.pyx reads: M[i_k, add]
                    have = true;
                }
            for (int c = 0; c < W && ok; ++c)
                for (int d = 0; d < W && ok; ++d) {
                    const int a = t->ewald_inds[(size_t)s * W + c], b = t->ewald_inds[(size_t)u * W + d];
                    if (a < 0 || b < 0) continue;
                    if (fabs(Mx[(size_t)b * M + a] - q[a] * q[b] * g) > tol) ok = false;
                }
            G[(size_t)s * N + u] = g; // row s: kernel between the flipped site s and site u
        }
    if (!ok) return 0; // not of product form: keep the dense rows
    // split sites into changeable ones and single-species ("frozen") ones
    std::vector<int> act;
    std::vector<char> frozen(N, 0);
    for (int s = 0; s < N; ++s) {
        int nvalid = 0, ncodes = 0;
        for (int c = 0; c < W; ++c) nvalid += t->ewald_inds[(size_t)s * W + c] >= 0;
        (void)ncodes;
        // a site is frozen when it is in no active sublattice (its code never changes) and
        // carries exactly one Ewald species (code 0)
        bool in_active = false;
        for (int64_t i = 0; i < t->sub_site_ptr[t->n_sublattices] && !in_active; ++i)
            in_active = t->sub_active_sites[i] == s;
        frozen[s] = (!in_active && nvalid == 1 && t->ewald_inds[(size_t)s * W] >= 0) ? 1 : 0;
        if (!frozen[s]) act.push_back(s);
    }
    const size_t na = act.size();
    std::vector<double> Gact((size_t)N * std::max<size_t>(na, 1), 0.0), fz(N, 0.0);
    for (int s = 0; s < N; ++s) {
        for (size_t j = 0; j < na; ++j) Gact[(size_t)s * na + j] = G[(size_t)s * N + act[j]];
        double c = 0;
        for (int u = 0; u < N; ++u)
            if (frozen[u] && u != s) c += qs[(size_t)u * W] * G[(size_t)s * N + u];
        fz[s] = c;
    }
    TRY(dev_upload(h, act.data(), act.size(), &h->kp.ew_act));
    TRY(dev_upload(h, fz.data(), fz.size(), &h->kp.ew_frozen));
    h->kp.ew_nact = (int)na;
    h->kp.ew_act_base = na ? act[0] : -1;
    for (size_t j = 0; j < na; ++j)
        if (act[j] != act[0] + (int)j) h->kp.ew_act_base = -1;
    G.swap(Gact);
    TRY(compress_ewald_rows(h, t, G, act));
    TRY(dev_upload(h, G.data(), G.size(), &h->kp.ew_G));
    TRY(dev_upload(h, qs.data(), qs.size(), &h->kp.ew_qs));
    TRY(dev_upload(h, dg.data(), dg.size(), &h->kp.ew_dg));
    h->ew_qs_host = qs;
    h->ew_dg_host = dg;
    h->kp.ew_compact = 1;
    return 0;
}

extern "C" int smolmc_abi_version(void) { return SMOLMC_ABI_VERSION; }
extern "C" const char *smolmc_last_error(void) { return smolmc_g_err.c_str(); }

// ---- site relabelling behind the boundary (ABI 8) -------------------------------------------------------------
// The lean kernel families index an active sublattice as ONE site range (site = base + mulhi(word, n_active)).
// Ensemble.restrict_sites and split_sublattice_by_species (smol/moca/sublattice.py:84-186, ensemble.py:288-321)
// leave the active sites of a sublattice scattered.  smolmc_create then renumbers the sites ITSELF: the active sites
// of every sublattice first, in the order of their lists (so a walker draws the same physical site from the same
// random word), then the other sites whose species can differ between occupancies (restricted sites: the Ewald
// potential field covers one block of changeable sites), then the rest, each group in the caller's order.  The handle
// is built from the renumbered tables; occupancies, step records and sample rows cross every entry point in the
// CALLER's numbering (set_state / get_state / get_samples* / replay / eval_full / eval_delta).  Rounds 4-5 did this in
// the Python binding only (capi.TableSet.permute_sites), so a C client of this header got mc_kernel.
struct RelabelledTables {
    smolmc_tables t;
    std::vector<int32_t> full_idx, loc_idx, sub_active_sites, loc_orbit, loc_nrows, ewald_inds;
    std::vector<int64_t> site_ptr, loc_off;
    std::vector<double> loc_ratio, mu_table, bias_table;
};

// new_of (caller's site -> engine's site) when a relabelling is needed and possible, else empty
static std::vector<int32_t> plan_relabelling(const smolmc_tables *t) {
    std::vector<int32_t> none;
    const int N = t->num_sites, ns = t->n_sublattices;
    if (N <= 0 || ns <= 0 || !t->sub_site_ptr || !t->sub_active_sites || getenv("SMOLMC_NO_SITE_RELABEL")) return none;
    bool needed = false;
    std::vector<char> taken((size_t)N, 0);
    std::vector<int32_t> order;
    order.reserve((size_t)N);
    for (int k = 0; k < ns; ++k) {
        const int64_t a = t->sub_site_ptr[k], b = t->sub_site_ptr[k + 1];
        for (int64_t i = a; i < b; ++i) {
            const int st = t->sub_active_sites[i];
            if (st < 0 || st >= N || taken[st]) return none; // (malformed lists: smolmc_create reports them)
            taken[st] = 1;
            order.push_back(st);
            if (st != t->sub_active_sites[a] + (int)(i - a)) needed = true;
        }
    }
    if (!needed) return none;
    // sites outside the active lists: changeable ones (more than one Ewald species, or a vacancy: what
    // build_compact_ewald does not fold into the frozen-site constants) before the single-species ones
    auto frozen = [&](int s) {
        if (!t->has_ewald || !t->ewald_inds) return true;
        int nvalid = 0;
        for (int c = 0; c < t->ewald_width; ++c) nvalid += t->ewald_inds[(size_t)s * t->ewald_width + c] >= 0;
        return nvalid == 1 && t->ewald_inds[(size_t)s * t->ewald_width] >= 0;
    };
    for (int pass = 0; pass < 2; ++pass)
        for (int s = 0; s < N; ++s)
            if (!taken[s] && frozen(s) == (pass == 1)) order.push_back(s);
    if ((int)order.size() != N) return none;
    std::vector<int32_t> new_of((size_t)N);
    for (int p = 0; p < N; ++p) new_of[order[p]] = p;
    return new_of;
}

// the caller's tables with site p renamed new_of[p] in every site-indexed array (validate_tables has passed)
static void relabel_tables(const smolmc_tables *t, const std::vector<int32_t> &new_of, const std::vector<int32_t> &old_of,
                           RelabelledTables &rt) {
    const int N = t->num_sites;
    rt.t = *t;
    const int64_t nfull = t->full_off[t->n_orb], nrec = t->site_ptr[N];
    rt.full_idx.resize((size_t)nfull);
    for (int64_t i = 0; i < nfull; ++i) rt.full_idx[i] = new_of[t->full_idx[i]];
    int64_t nloc = 0;
    for (int64_t r = 0; r < nrec; ++r)
        nloc = std::max<int64_t>(nloc, t->loc_off[r] + (int64_t)t->loc_nrows[r] * t->orb_nsites[t->loc_orbit[r]]);
    rt.loc_idx.resize((size_t)nloc);
    for (int64_t i = 0; i < nloc; ++i) {
        const int v = t->loc_idx[i];
        rt.loc_idx[i] = (v >= 0 && v < N) ? new_of[v] : v; // (entries between records, if any, are never read)
    }
    // the record range of engine site p is that of the caller's site old_of[p]
    rt.site_ptr.assign((size_t)N + 1, 0);
    rt.loc_orbit.reserve((size_t)nrec); rt.loc_nrows.reserve((size_t)nrec);
    rt.loc_off.reserve((size_t)nrec); rt.loc_ratio.reserve((size_t)nrec);
    for (int p = 0; p < N; ++p) {
        const int q = old_of[p];
        for (int64_t r = t->site_ptr[q]; r < t->site_ptr[q + 1]; ++r) {
            rt.loc_orbit.push_back(t->loc_orbit[r]);
            rt.loc_nrows.push_back(t->loc_nrows[r]);
            rt.loc_off.push_back(t->loc_off[r]);
            rt.loc_ratio.push_back(t->loc_ratio[r]);
        }
        rt.site_ptr[(size_t)p + 1] = (int64_t)rt.loc_orbit.size();
    }
    const int64_t nact = t->sub_site_ptr[t->n_sublattices];
    rt.sub_active_sites.resize((size_t)nact);
    for (int64_t i = 0; i < nact; ++i) rt.sub_active_sites[i] = new_of[t->sub_active_sites[i]];
    auto rows = [&](const auto *src, int width, int blocks, auto &dst) {
        dst.resize((size_t)blocks * N * width);
        for (int b = 0; b < blocks; ++b)
            for (int p = 0; p < N; ++p)
                std::copy(src + ((size_t)b * N + old_of[p]) * width, src + ((size_t)b * N + old_of[p] + 1) * width,
                          dst.begin() + ((size_t)b * N + p) * width);
    };
    rt.t.full_idx = rt.full_idx.data();
    rt.t.loc_idx = rt.loc_idx.data();
    rt.t.site_ptr = rt.site_ptr.data();
    rt.t.loc_orbit = rt.loc_orbit.data();
    rt.t.loc_nrows = rt.loc_nrows.data();
    rt.t.loc_off = rt.loc_off.data();
    rt.t.loc_ratio = rt.loc_ratio.data();
    rt.t.sub_active_sites = rt.sub_active_sites.data();
    if (t->has_ewald && t->ewald_inds) {
        rows(t->ewald_inds, t->ewald_width, 1, rt.ewald_inds);
        rt.t.ewald_inds = rt.ewald_inds.data();
    }
    if (t->has_mu && t->mu_table) {
        rows(t->mu_table, t->mu_width, 1, rt.mu_table);
        rt.t.mu_table = rt.mu_table.data();
    }
    if (t->bias_type && t->bias_table && t->bias_width > 0) {
        const int brows = t->bias_type == SMOLMC_BIAS_SQUARE_HYPERPLANE ? std::max(1, std::min(t->bias_rows, SMOLMC_MAX_BIAS_ROWS)) : 1;
        rows(t->bias_table, t->bias_width, brows, rt.bias_table);
        rt.t.bias_table = rt.bias_table.data();
    }
}

static int create_impl(const smolmc_tables *t, const smolmc_config *cfg, smolmc_handle **out, std::vector<int32_t> *new_of);

// Test hooks, NOT part of the C-ABI (not in include/smolmc.h): the renumbering smolmc_create would apply to these
// tables, as a heap copy whose first member is the renumbered smolmc_tables (NULL: none needed / possible), and the
// map caller's site -> engine's site.  Host code only: tests/test_relabel_host.py runs the CPU oracle on both table
// sets without a GPU.
extern "C" void *smolmc_debug_relabel(const smolmc_tables *t, int32_t *new_of_out) {
    if (!t || validate_tables(t)) return nullptr;
    std::vector<int32_t> new_of = plan_relabelling(t);
    if (new_of.empty()) return nullptr;
    std::vector<int32_t> old_of(new_of.size());
    for (size_t p = 0; p < new_of.size(); ++p) old_of[new_of[p]] = (int32_t)p;
    RelabelledTables *rt = new RelabelledTables();
    relabel_tables(t, new_of, old_of, *rt);
    if (new_of_out) std::copy(new_of.begin(), new_of.end(), new_of_out);
    return rt;
}
extern "C" void smolmc_debug_relabel_free(void *p) { delete (RelabelledTables *)p; }

extern "C" int smolmc_create(const smolmc_tables *t, const smolmc_config *cfg, smolmc_handle **out) {
    if (!t || !cfg || !out) return fail("null argument");
    if (cfg->n_replicas <= 0) return fail("n_replicas must be positive");
    if (int rc = validate_tables(t)) return rc; // (in the caller's numbering, before anything is renamed)
    std::vector<int32_t> new_of = plan_relabelling(t);
    if (new_of.empty()) return create_impl(t, cfg, out, nullptr);
    std::vector<int32_t> old_of(new_of.size());
    for (size_t p = 0; p < new_of.size(); ++p) old_of[new_of[p]] = (int32_t)p;
    RelabelledTables rt;
    relabel_tables(t, new_of, old_of, rt);
    return create_impl(&rt.t, cfg, out, &new_of); // (the engine copies every table at create: rt may go)
}

static int create_impl(const smolmc_tables *t, const smolmc_config *cfg, smolmc_handle **out, std::vector<int32_t> *new_of) {
    if (t->max_species > 255) return fail("more than 255 species codes per site");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("no HIP device available: the smol_amd engine requires an AMD GPU (there is no "
                    "CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("invalid device ordinal");
    smolmc_handle *h = new smolmc_handle();
    h->cfg = *cfg;
    h->device = cfg->device;
    if (new_of) {
        h->relabelled = true;
        h->new_of = *new_of;
        h->old_of.resize(new_of->size());
        for (size_t p = 0; p < new_of->size(); ++p) h->old_of[(*new_of)[p]] = (int32_t)p;
    }
    auto bail = [&](int rc) {
        smolmc_destroy(h);
        return rc;
    };
    if (hipSetDevice(h->device) != hipSuccess) return bail(fail("hipSetDevice failed"));
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail("hipStreamCreate failed"));
    h->own_stream = true;
    hipEventCreate(&h->ev0);
    hipEventCreate(&h->ev1);
    h->R = cfg->n_replicas;
    h->N = t->num_sites;
    h->Npad = (t->num_sites + 15) / 16 * 16;
    h->Fce = num_ce_features(t);
    h->F = h->Fce + (t->has_ewald ? 1 : 0) + (t->has_mu ? 1 : 0);
    h->natural.assign(t->ce_coefs, t->ce_coefs + h->Fce);
    if (t->has_ewald) h->natural.push_back(t->ewald_coef);
    if (t->has_mu) h->natural.push_back(-1.0);
    const bool wl = cfg->kernel_type == SMOLMC_KERNEL_WANGLANDAU;
    if (wl) {
        if (cfg->wl_min_enthalpy > cfg->wl_max_enthalpy)
            return bail(fail("min_enthalpy can not be larger than max_enthalpy.")); // wanglandau.py:82
        if ((cfg->wl_max_enthalpy - cfg->wl_min_enthalpy) / cfg->wl_bin_size <= 1)
            return bail(fail("The values provided for min and max enthalpy and bin sizer result in a "
                             "single bin!"));
        if (cfg->wl_mod_factor <= 0) return bail(fail("mod_factor must be greater than 0."));
        h->L = (int)ceil((cfg->wl_max_enthalpy - cfg->wl_min_enthalpy) / cfg->wl_bin_size);
    }
    memset(&h->kp, 0, sizeof(KParams));
    KParams &kp = h->kp;
    // (validate_tables(t) has passed: smolmc_create runs it on the caller's tables before any renaming)
    // species codes a site may carry: max_species by default, the largest sublattice code + 1 on
    // the sites of an active sublattice
    h->site_ncodes.assign((size_t)t->num_sites, (uint8_t)std::min(255, std::max(1, t->max_species)));
    h->site_active.assign((size_t)t->num_sites, 0);
    for (int k = 0; k < t->n_sublattices; ++k) {
        int top = 0;
        for (int64_t c = t->sub_code_ptr[k]; c < t->sub_code_ptr[k + 1]; ++c) top = std::max(top, t->sub_codes[c]);
        for (int64_t i = t->sub_site_ptr[k]; i < t->sub_site_ptr[k + 1]; ++i) {
            const int st = t->sub_active_sites[i];
            if (st >= 0 && st < t->num_sites) {
                h->site_ncodes[st] = (uint8_t)std::min(255, top + 1);
                h->site_active[st] = 1;
            }
        }
    }
    // ... and on every other site the width of its site space as the cluster tensors see it: member m of
    // an orbit's rows has prev_stride / stride species (see validate_tables); a code beyond that would
    // index past the tensors
    {
        std::vector<int> seen((size_t)t->num_sites, 0);
        for (int o = 0; o < t->n_orb; ++o) {
            const int I = t->orb_nsites[o];
            const int32_t *st = t->tensor_indices + t->orb_stride_off[o];
            for (int64_t i = t->full_off[o]; i < t->full_off[o + 1]; ++i) {
                const int m = (int)((i - t->full_off[o]) % I);
                const int width = (int)((m == 0 ? t->orb_tensor_len[o] : st[m - 1]) / st[m]);
                const int site = t->full_idx[i];
                seen[site] = seen[site] ? std::min(seen[site], width) : width;
            }
        }
        for (int s = 0; s < t->num_sites; ++s)
            if (!h->site_active[s] && seen[s]) h->site_ncodes[s] = (uint8_t)std::min<int>(h->site_ncodes[s], std::min(255, seen[s]));
    }
    if (int rc = build_mc_tables(h, t)) return bail(rc);
    if (int rc = build_ref_tables(h, t)) return bail(rc);
    kp.N = h->N;
    kp.Npad = h->Npad;
    kp.Fce = h->Fce;
    kp.F = h->F;
    kp.step_type = cfg->step_type;
    kp.corr_mode = h->rt.corr_mode;
    kp.has_ewald = t->has_ewald;
    kp.has_mu = t->has_mu;
    kp.ew_W = h->rt.ew_W;
    kp.ew_M = h->rt.ew_M;
    kp.mu_W = h->rt.mu_W;
    kp.ew_inds = h->rt.ew_inds;
    kp.ew_Mt = h->rt.ew_Mt;
    kp.ew_coef = t->ewald_coef;
    kp.mu = h->rt.mu;
    // MCBias (smol/moca/kernel/bias.py)
    if (t->bias_type) {
        if (t->bias_type != SMOLMC_BIAS_FUGACITY && t->bias_type != SMOLMC_BIAS_SQUARE_CHARGE &&
            t->bias_type != SMOLMC_BIAS_SQUARE_HYPERPLANE)
            return bail(fail("unknown bias_type"));
        const int brows = t->bias_type == SMOLMC_BIAS_SQUARE_HYPERPLANE ? t->bias_rows : 1;
        if (brows < 1 || brows > SMOLMC_MAX_BIAS_ROWS)
            return bail(fail("bias_rows must be in [1, SMOLMC_MAX_BIAS_ROWS]"));
        if (wl) return bail(fail("Cannot apply bias to Wang-Landau simulation!")); // wanglandau.py:127-128
        if (!t->bias_table || t->bias_width < t->max_species)
            return bail(fail("bias_table must be [num_sites x >= max_species]"));
        if (t->bias_type != SMOLMC_BIAS_FUGACITY && !(t->bias_penalty > 0.0))
            return bail(fail("Penalty factor should be > 0!")); // bias.py:250-251, :328-329
        h->bias_host.assign(t->bias_table, t->bias_table + (size_t)brows * t->num_sites * t->bias_width);
        h->bias_icpt.assign(SMOLMC_MAX_BIAS_ROWS, 0.0);
        if (t->bias_type == SMOLMC_BIAS_SQUARE_HYPERPLANE && t->bias_intercepts)
            for (int k = 0; k < brows; ++k) h->bias_icpt[k] = t->bias_intercepts[k];
        kp.bias_rows = brows;
        kp.bias_row_stride = (size_t)t->num_sites * t->bias_width;
        if (t->bias_type == SMOLMC_BIAS_FUGACITY)
            for (double v : h->bias_host)
                if (!(v > 0.0)) return bail(fail("fugacity fractions must be positive"));
        kp.bias_type = t->bias_type;
        kp.bias_W = t->bias_width;
        kp.bias_pen = t->bias_penalty;
        if (dev_upload(h, h->bias_host.data(), h->bias_host.size(), &kp.bias_tab)) return bail(1);
        if (dev_alloc(h, (size_t)cfg->n_replicas, &kp.bias) ||
            dev_alloc(h, (size_t)cfg->n_replicas * SMOLMC_MAX_BIAS_ROWS, &kp.charge))
            return bail(1);
    }
    if (t->has_ewald && t->ewald_charges && getenv("SMOLMC_DENSE_EWALD") == nullptr)
        if (int rc = build_compact_ewald(h, t)) return bail(rc);
    // potential field of every walker in HBM: needs the compact form and
    // contiguous changeable sites; the lean kernels stage the same buffer through LDS
    kp.ew_field = 0;
    if (kp.ew_compact && kp.ew_act_base >= 0 && (size_t)kp.ew_nact * 8 <= 150 * 1024 &&
        getenv("SMOLMC_NO_EWALD_FIELD") == nullptr) {
        if (dev_alloc(h, (size_t)cfg->n_replicas * kp.ew_nact, &kp.ew_phi)) return bail(1);
        kp.ew_field = 1;
    }
    // sublattices
    {
        const int ns = t->n_sublattices;
        if (ns <= 0) return bail(fail("no active sublattice"));
        std::vector<int> ptr(ns + 1), cptr(ns + 1), base(ns);
        std::vector<double> cum(ns);
        double c = 0;
        for (int s = 0; s <= ns; ++s) {
            ptr[s] = (int)t->sub_site_ptr[s];
            cptr[s] = (int)t->sub_code_ptr[s];
        }
        for (int s = 0; s < ns; ++s) {
            c += t->sub_probs[s];
            cum[s] = c;
            const int a = ptr[s], b = ptr[s + 1];
            if (b <= a) return bail(fail("empty active sublattice"));
            bool contig = true;
            for (int i = a + 1; i < b; ++i)
                if (t->sub_active_sites[i] != t->sub_active_sites[a] + (i - a)) contig = false;
            base[s] = contig ? t->sub_active_sites[a] : -1;
            for (int i = a; i < b; ++i)
                if (t->sub_active_sites[i] < 0 || t->sub_active_sites[i] >= t->num_sites)
                    return bail(fail("sublattice site index out of range"));
        }
        kp.nsub = ns;
        if (dev_upload(h, ptr.data(), ptr.size(), &kp.sub_ptr) ||
            dev_upload(h, t->sub_active_sites, (size_t)ptr[ns], &kp.sub_sites) ||
            dev_upload(h, base.data(), base.size(), &kp.sub_base) ||
            dev_upload(h, cptr.data(), cptr.size(), &kp.sub_code_ptr) ||
            dev_upload(h, t->sub_codes, (size_t)cptr[ns], &kp.sub_codes) ||
            dev_upload(h, cum.data(), cum.size(), &kp.sub_cum))
            return bail(1);
    }
    // walker state
    const size_t R = h->R;
    kp.R = h->R;
    double *dbeta = nullptr;
    uint64_t *dseeds = nullptr;
    if (dev_alloc(h, R * h->Npad, &kp.occ) || dev_alloc(h, R, &kp.enthalpy) ||
        dev_alloc(h, R * h->F, &kp.features) || dev_alloc(h, R, &dbeta) || dev_alloc(h, R, &dseeds) ||
        dev_alloc(h, R, &kp.nsteps) || dev_alloc(h, R, &kp.nacc) || dev_alloc(h, R, &kp.last_acc))
        return bail(1);
    kp.beta = dbeta;
    h->d_beta = dbeta;
    kp.seeds = dseeds;
    if (dev_upload(h, h->natural.data(), h->natural.size(), (const double **)&h->d_natural))
        return bail(1);
    if (wl) {
        kp.L = h->L;
        kp.wl_min = cfg->wl_min_enthalpy;
        kp.wl_max = cfg->wl_max_enthalpy;
        kp.wl_bin = cfg->wl_bin_size;
        kp.wl_flat = cfg->wl_flatness;
        kp.wl_div = cfg->wl_mod_divisor;
        kp.wl_check = cfg->wl_check_period;
        kp.wl_update = cfg->wl_update_period;
        kp.wl_sum_mode = (cfg->wl_update_period == 1 && getenv("SMOLMC_WL_RUNNING_MEAN") == nullptr) ? 1 : 0;
        // (check period 0: the flatness check is the caller's -- a host-side mod_update callable,
        // wanglandau.py:100-105 -- and no kernel runs its own)
        if (kp.wl_check < 0 || kp.wl_update <= 0) return bail(fail("WL periods must be positive"));
        // (the lane-indexed feature bookkeeping of mc_kernel / mc_wl_kernel holds 64 features; more -- the
        // reference has no limit, wanglandau.py:117-118 -- take the universal kernel)
        if (h->F > 64 && h->general_ok) no_general(h, "Wang-Landau with more than 64 features");
        if (dev_alloc(h, R * h->L, &kp.wl_entropy) || dev_alloc(h, R * h->L, &kp.wl_hist) ||
            dev_alloc(h, R * h->L, &kp.wl_occur) || dev_alloc(h, R * h->L * h->F + 64, &kp.wl_meanf) || // (+64: the lean kernel's all-lane atomic, see mc_lean.h)
            dev_alloc(h, R, &kp.wl_m) || dev_alloc(h, R, &kp.wl_counter))
            return bail(1);
        std::vector<double> m0(R, cfg->wl_mod_factor);
        if (hipMemcpy(kp.wl_m, m0.data(), R * 8, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail("hipMemcpy failed"));
    }
    // LDS layout
    size_t tb = (size_t)kp.nclasses * kp.Cpad * (16 + 16 + 8) + (size_t)(kp.xt_len + kp.ft_len) * 8 +
                (size_t)((kp.nclasses + 3) & ~3) * 4 + (kp.nclasses > 1 ? (size_t)h->N : 0);
    tb = (tb + 15) / 16 * 16;
    size_t pw = (size_t)h->Npad;
    if (wl)
        pw += (size_t)h->L * 24 + (size_t)h->F * 8;
    else {
        const size_t by_feat = (size_t)h->Fce * 64 * 8;
        const size_t by_slot = ((size_t)kp.nclasses * kp.Cpad + h->Fce) * 8;
        kp.acc_by_slot = (!kp.corr_mode && by_slot < by_feat) ? 1 : 0;
        pw += kp.acc_by_slot ? by_slot : by_feat;
    }
    pw = (pw + 15) / 16 * 16;
    kp.lds_tables = (int)tb;
    kp.lds_per_wave = (int)pw;
    h->waves_per_block = 4;
    while (h->waves_per_block > 1 && tb + pw * h->waves_per_block > 160 * 1024) h->waves_per_block /= 2;
    h->lds_bytes = tb + pw * h->waves_per_block;
    if (h->general_ok && h->lds_bytes > 160 * 1024)
        no_general(h, "tables + one chain exceed the 160 KiB LDS budget of mc_kernel");
    // lean-kernel eligibility (everything else runs mc_kernel)
    {
        // (lean Wang-Landau keeps per-bin feature SUMS: update_period 1 only, see WlParams)
        // (lazy cluster features: the lean kernels see the scalar features only -- Ewald energy, chemical work)
        const int nscal = (t->has_ewald ? 1 : 0) + (t->has_mu ? 1 : 0);
        const int Fk = h->lazy_tables ? nscal : h->F;
        if (h->lazy_tables) {
            if (dev_alloc(h, (size_t)h->R * 2, &h->d_lazy_scal)) return bail(1);
        }
        // (... the Wang-Landau TableFlip kernel, mc_table_kernel<..., WLT>, also keeps running means: any update_period)
        const bool table_wl = wl && cfg->step_type == SMOLMC_STEP_TABLE_FLIP;
        bool lean = h->lean_tables && Fk <= 64 &&
                    (!wl || (((cfg->wl_update_period == 1 && getenv("SMOLMC_WL_RUNNING_MEAN") == nullptr) || (table_wl && cfg->wl_update_period < (1ll << 31))) &&
                             h->F <= 63 && cfg->wl_check_period < (1ll << 31) && // (cell 63 of the feature scratch is the kernel's zero)
                             (getenv("SMOLMC_WL_PLAIN_ONLY") == nullptr || (!t->has_ewald && !t->has_mu)))) &&
                    (!t->has_ewald || kp.ew_compact) && t->n_sublattices == 1 && h->lean_ncls == 1 &&
                    h->lean_nslot <= 4 && getenv("SMOLMC_FORCE_GENERAL") == nullptr;
        int sbase = -1, nact = 0, nc = 0;
        std::vector<double> mu_row;
        if (lean) {
            nact = (int)(t->sub_site_ptr[1] - t->sub_site_ptr[0]);
            nc = (int)(t->sub_code_ptr[1] - t->sub_code_ptr[0]);
            sbase = t->sub_active_sites[0];
            for (int i = 0; i < nact; ++i)
                if (t->sub_active_sites[i] != sbase + i) lean = false;
            // Species codes 0 .. n-1, or -- canonical swaps only, which never draw a code: they exchange the two sites'
            // own -- any code list (a sublattice split by species, sublattice.py:109-186, keeps the site space's codes:
            // {1, 2} of three); the per-code rows (mu, charges, bias pairs) then run up to the largest code.
            bool default_codes = true;
            int top = 0;
            for (int c = 0; c < nc; ++c) {
                default_codes = default_codes && t->sub_codes[c] == c;
                top = std::max(top, (int)t->sub_codes[c]);
            }
            if (nc < 2 || nc > 8) lean = false;
            if (!default_codes) {
                if (cfg->step_type == SMOLMC_STEP_SWAP && top < 8 && getenv("SMOLMC_NO_SWAP_ANY_CODES") == nullptr) nc = top + 1;
                else lean = false;
            }
        }
        if (lean && t->has_mu) {
            if (t->mu_width < nc) lean = false;
            for (int c = 0; lean && c < nc; ++c) mu_row.push_back(t->mu_table[(size_t)sbase * t->mu_width + c]);
            for (int i = 0; lean && i < nact; ++i)
                for (int c = 0; c < nc; ++c)
                    if (t->mu_table[(size_t)(sbase + i) * t->mu_width + c] != mu_row[c]) lean = false;
        }
        // (pair tables of the bias: [row][old * 8 + new], one row for Fugacity / SquareCharge, bias_rows rows for
        // SquareHyperplaneBias -- on the lean kernels since round 5)
        const int brows_l = t->bias_type == SMOLMC_BIAS_SQUARE_HYPERPLANE ? t->bias_rows : 1;
        std::vector<double> bias_pair((size_t)64 * SMOLMC_MAX_BIAS_ROWS, 0.0);
        // the table kernels take at most 8 flip vectors: larger tables run on the universal kernel
        // (Wang-Landau TableFlip: mc_table_kernel<..., WLT> since round 6; SMOLMC_NO_TABLE_WL: A/B switch)
        if (lean && cfg->step_type == SMOLMC_STEP_TABLE_FLIP && (t->n_flip_vectors > 8 || (wl && getenv("SMOLMC_NO_TABLE_WL") != nullptr)))
            lean = false;
        // several correlation functions per orbit: plain Metropolis flip / swap variants only
        if (lean && h->lean_kf && (wl || t->bias_type || cfg->step_type == SMOLMC_STEP_TABLE_FLIP)) lean = false;
        if (lean && t->bias_type) {
            // the bias row must be the same on every active site (it is defined per sublattice)
            const int W = t->bias_width;
            for (int k = 0; k < brows_l; ++k) {
                const double *tabk = h->bias_host.data() + (size_t)k * t->num_sites * W;
                for (int i = 0; lean && i < nact; ++i)
                    for (int c = 0; c < nc; ++c)
                        if (tabk[(size_t)(sbase + i) * W + c] != tabk[(size_t)sbase * W + c]) lean = false;
                for (int o = 0; lean && o < nc; ++o)
                    for (int n = 0; n < nc; ++n) {
                        const double a = tabk[(size_t)sbase * W + n], b = tabk[(size_t)sbase * W + o];
                        bias_pair[(size_t)k * 64 + o * 8 + n] = t->bias_type == SMOLMC_BIAS_FUGACITY ? std::log(a / b) : a - b;
                    }
            }
            // (TableFlip with a bias: mc_table_kernel<..., BIAS> since round 6; SMOLMC_NO_TABLE_BIAS: A/B switch)
            if (cfg->step_type == SMOLMC_STEP_TABLE_FLIP && getenv("SMOLMC_NO_TABLE_BIAS") != nullptr) lean = false;
        }
        if (lean) {
            LeanParams &lp = h->lp;
            if (t->has_mu && dev_upload(h, mu_row.data(), mu_row.size(), &lp.mu_row)) return bail(1);
            if (t->bias_type) {
                if (dev_upload(h, bias_pair.data(), bias_pair.size(), &lp.bias_pair)) return bail(1);
                lp.bias_type = t->bias_type;
                lp.bias_pen = t->bias_penalty;
                lp.bias = kp.bias;
                lp.charge = kp.charge;
                lp.bias_rows = brows_l;
                lp.bias_row_stride = 64;
            }
            if (t->has_mu) { // the mu delta of a step joins the float32 sum (<= 2 flips * 2 |mu|)
                double mmax = 0.0;
                for (double v : mu_row) mmax = std::max(mmax, std::fabs(v));
                lp.fast_eps += 4.0 * mmax * ldexp(1.0, -19);
            }
            if (getenv("SMOLMC_NO_FAST_ACCEPT")) lp.fast_eps = 0.0; // A/B switch
            // test hook: widen the undecided band so that both decision paths interleave
            if (const char *sc = getenv("SMOLMC_FAST_EPS_SCALE")) lp.fast_eps *= atof(sc);
            lp.occ = kp.occ;
            lp.enthalpy = kp.enthalpy;
            lp.features = h->lazy_tables ? h->d_lazy_scal : kp.features;
            lp.beta = kp.beta;
            lp.seeds = kp.seeds;
            lp.nsteps = kp.nsteps;
            lp.nacc = kp.nacc;
            lp.last_acc = kp.last_acc;
            lp.R = h->R;
            lp.N = h->N;
            lp.Npad = h->Npad;
            lp.F = Fk;
            lp.Fce = h->lazy_tables ? 0 : h->Fce;
            lp.sbase = sbase;
            lp.nact = nact;
            lp.ncodes = nc;
            if (t->has_ewald) {
                lp.ew_W = kp.ew_W;
                lp.ew_nact = kp.ew_nact;
                lp.ew_act_base = kp.ew_act_base;
                lp.ew_act = kp.ew_act;
                lp.ew_G = kp.ew_G;
                lp.ew_qs = kp.ew_qs;
                lp.ew_dg = kp.ew_dg;
                lp.ew_frozen = kp.ew_frozen;
                lp.ew_coef = kp.ew_coef;
                lp.ew_gx = kp.ew_gx; lp.ew_E8 = kp.ew_E8; lp.ew_S8 = kp.ew_S8;
            }
            if (cfg->step_type == SMOLMC_STEP_TABLE_FLIP) {
                if (dev_upload(h, t->flip_table, (size_t)t->n_flip_vectors * nc, &lp.tf_table) ||
                    dev_upload(h, t->flip_weights, (size_t)2 * t->n_flip_vectors, &lp.tf_w))
                    return bail(1);
                lp.tf_n = t->n_flip_vectors;
                lp.tf_sw = t->swap_weight;
            }
            if (wl) {
                lp.wl.L = h->L;
                lp.wl.vmin = kp.wl_min; lp.wl.vmax = kp.wl_max; lp.wl.bin = kp.wl_bin;
                lp.wl.flat = kp.wl_flat; lp.wl.div = kp.wl_div;
                lp.wl.check = kp.wl_check; lp.wl.update = kp.wl_update;
                lp.wl.entropy = kp.wl_entropy; lp.wl.hist = kp.wl_hist; lp.wl.occur = kp.wl_occur;
                lp.wl.meanf = kp.wl_meanf; lp.wl.m = kp.wl_m; lp.wl.counter = kp.wl_counter;
                lp.wl.sum_mode = table_wl ? kp.wl_sum_mode : 1;
            }
            // (Wang-Landau: per-bin records and the cached rows of per-bin feature sums, mc_wl.h)
            h->lean_lds = ((size_t)lp.dt_len + 24) * 8 +
                          (size_t)4 * (lp.Nlds + 64 * 8 + 64 +
                                       (table_wl ? wl_multi_wave_bytes(h->L, h->F, kp.wl_sum_mode) // (mc_table_kernel<..., WLT>: the multi-class kernel's state)
                                        : wl     ? std::max(wl_lean_bins_bytes(h->L) + (size_t)SMOLMC_WL_ROWS * h->F * 8,
                                                            // (round 2's variant inside mc_lean_kernel, A/B switch: 24-byte records)
                                                            getenv("SMOLMC_WL_V2") ? (size_t)h->L * 24 : (size_t)0)
                                                 : 0));
            if (h->lean_lds > 150 * 1024) lean = false;
            // Ewald potential field in LDS when the changeable sites are the active
            // sublattice and it fits beside the occupancies
            lp.ew_field = 0;
            // (the changeable sites may extend beyond the active ones: restricted sites, which the host
            // API relabels behind the active sites of their sublattice; the field covers them all)
            const int nfield = kp.ew_nact;
            if (lean && t->has_ewald && kp.ew_act_base == sbase && nfield >= nact &&
                (nfield == nact || getenv("SMOLMC_FIELD_ACTIVE_ONLY") == nullptr) &&
                getenv("SMOLMC_NO_EWALD_FIELD") == nullptr) {
                const size_t with_field = h->lean_lds + (size_t)4 * nfield * 8;
                // charge / diagonal term per species code must not depend on the site
                bool uniform = kp.ew_W <= 8;
                std::vector<double> qrow(8, 0.0), dgrow(8, 0.0);
                for (int c = 0; uniform && c < kp.ew_W; ++c) {
                    qrow[c] = h->ew_qs_host[(size_t)sbase * kp.ew_W + c];
                    dgrow[c] = h->ew_dg_host[(size_t)sbase * kp.ew_W + c];
                    for (int i = 0; i < nfield; ++i)
                        if (h->ew_qs_host[(size_t)(sbase + i) * kp.ew_W + c] != qrow[c] ||
                            h->ew_dg_host[(size_t)(sbase + i) * kp.ew_W + c] != dgrow[c])
                            uniform = false;
                }
                if (uniform && kp.ew_field && with_field <= 150 * 1024 && (size_t)nfield * 8 <= 64 * 1024) {
                    if (dev_upload(h, qrow.data(), 8, &lp.ew_qrow) || dev_upload(h, dgrow.data(), 8, &lp.ew_dgrow))
                        return bail(1);
                    lp.ew_phi = kp.ew_phi;
                    lp.ew_field = 1;
                    h->lean_lds = with_field;
                }
            }
            // Wang-Landau with the Ewald term: mc_wl_kernel takes it from the field in LDS only
            // (also what the round-2 kernel SMOLMC_WL_V2 cannot do: no Ewald, no mu)
            if (lean && wl && t->has_ewald && !lp.ew_field) lean = false;
            if (lean && wl && (t->has_ewald || t->has_mu || table_wl) && getenv("SMOLMC_WL_V2") != nullptr) lean = false;
            // one wave per workgroup (occupancy at LDS address 0, 32-bit index rows, a private
            // copy of the tables): Metropolis flips / swaps without Ewald term or bias, when 16
            // such workgroups still fit a CU
            if (lean && !wl && !t->has_ewald && !t->bias_type && !h->lean_kf && cfg->step_type != SMOLMC_STEP_TABLE_FLIP &&
                getenv("SMOLMC_NO_SOLO") == nullptr) {
                const size_t pw = ((size_t)lp.Nlds + 64 * 8 + 15) & ~(size_t)15;
                // (SMOLMC_EXP_F32TAB: room for the float32 shadow tables of the -DSMOLMC_EXP_F32TAB experiment build)
                const size_t solo_lds = pw + ((size_t)lp.dt_len + 24) * 8 +
                                        (getenv("SMOLMC_EXP_F32TAB") ? (((size_t)lp.dt_len * 4 + 15) & ~(size_t)15) : 0);
                if (solo_lds * 16 <= 160 * 1024 - 16 * 256) {
                    std::vector<uint32_t> wide(h->lean_idx_host.begin(), h->lean_idx_host.end());
                    if (dev_upload(h, wide.data(), wide.size(), &lp.idx32)) return bail(1);
                    h->lean_solo = true;
                    h->lean_lds = solo_lds;
                    // More walkers than the default register allocation (4 waves per SIMD) keeps
                    // resident: the 6-waves-per-SIMD instantiation, when 24 workgroups fit a CU.
                    // Measured (headline model, 1e4 steps): 6144 walkers 8.74 -> 7.08 ms, 16384
                    // walkers 19.7 -> 17.6 ms; at <= 4 waves per SIMD it is 7 % slower.
                    int cus = 0;
                    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 256;
                    if ((long)cfg->n_replicas > 16L * cus && solo_lds * 24 <= 160 * 1024 - 24 * 256 &&
                        getenv("SMOLMC_NO_OCC6") == nullptr)
                        h->lean_occ = 6;
                }
            }
            if (lean && cfg->step_type == SMOLMC_STEP_TABLE_FLIP) {
                // block-shared flip table / weights (48 doubles) and, when it fits, ln(k) for
                // k <= n_active from the host libm (the oracle's log-factorial sums use the same)
                h->lean_lds += 48 * 8;
                lp.tf_ln_len = 0;
                // ... unless it would cost the second resident workgroup per CU (160 KiB / 2)
                const size_t half = 80 * 1024, with_ln = h->lean_lds + (size_t)(nact + 1) * 8;
                std::vector<double> ln((size_t)nact + 1, 0.0);
                for (int k = 1; k <= nact; ++k) ln[k] = std::log((double)k);
                if (dev_upload(h, ln.data(), ln.size(), &lp.tf_ln)) return bail(1);
                if (with_ln <= 150 * 1024 && (with_ln <= half || h->lean_lds > half)) {
                    lp.tf_ln_len = nact + 1;
                    h->lean_lds += (size_t)(nact + 1) * 8;
                }
                // Eight walkers per workgroup when two four-walker workgroups would share a CU
                // anyway: waves w and w + 4 of a workgroup sit on the same SIMD, so that the
                // launch order (update_walker_order) can pair a hot walker with a cold one there.
                {
                    const size_t per_wave = (size_t)lp.Nlds + 64 * 8 + 64 + (lp.ew_field ? (size_t)kp.ew_nact * 8 : 0);
                    const size_t lds8 = h->lean_lds + 4 * per_wave;
                    h->lean_wpb = 4;
                    int cus = 0; // (fewer walkers than 8 per CU: four-walker workgroups spread over more CUs)
                    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 256;
                    if (!wl && lds8 <= 160 * 1024 && h->lean_lds > 40 * 1024 && cfg->n_replicas % 8 == 0 &&
                        (long)cfg->n_replicas >= 8L * cus && getenv("SMOLMC_TABLE_WPB4") == nullptr) {
                        h->lean_wpb = 8;
                        h->lean_lds_wpb8 = lds8;
                    }
                }
            }
        }
        h->lean = lean;
        // ---- several classes / sublattices (or > 256 clusters per site): mc_lean_multi_kernel
        const bool table = cfg->step_type == SMOLMC_STEP_TABLE_FLIP;
        // (MCBias: every bias type with flips or swaps, and since round 6 with TableFlip: mc_table_multi_kernel<..., BIAS>;
        // SMOLMC_NO_TABLE_BIAS: A/B switch)
        const bool multi_bias_ok = !t->bias_type || cfg->step_type != SMOLMC_STEP_TABLE_FLIP || getenv("SMOLMC_NO_TABLE_BIAS") == nullptr;
        // Wang-Landau on this layout (round 5; mc_lean_multi_kernel<..., WLK>): any number of classes the
        // layout takes and any update_period -- what mc_wl_kernel (one class, update_period 1) leaves.  The
        // Wang-Landau TableFlip: mc_table_multi_kernel<..., WLT> (round 6) with one correlation function per orbit, the
        // universal kernel otherwise.  SMOLMC_NO_WL_MULTI, SMOLMC_NO_TABLE_WL: A/B switches.
        const int wl_sum_mode = (cfg->wl_update_period == 1 && getenv("SMOLMC_WL_RUNNING_MEAN") == nullptr) ? 1 : 0;
        const bool multi_table_wl_ok = !h->lean_kf && getenv("SMOLMC_NO_TABLE_WL") == nullptr;
        const bool multi_wl_ok = !wl || ((!table || multi_table_wl_ok) && h->F <= 63 && cfg->wl_check_period < (1ll << 31) &&
                                         cfg->wl_update_period < (1ll << 31) && getenv("SMOLMC_NO_WL_MULTI") == nullptr);
        // (why a model with lean tables runs neither lean family: the first condition that fails, for smolmc_kernel_info)
        if (!lean && h->lean_tables)
            h->lean_reason = (h->lean_kf && !wl) ? "several correlation functions per orbit (KF kernel) on a model outside the single-class lean shape"
                             : !multi_wl_ok ? "Wang-Landau with more than 63 features, or with TableFlip and several correlation functions per orbit"
                             : !multi_bias_ok ? "environment override (SMOLMC_NO_TABLE_BIAS)"
                             : Fk > 64 ? "more than 64 features"
                             : t->n_sublattices > 4 ? "more than 4 active sublattices"
                             : (t->has_ewald && !kp.ew_field) ? "Ewald matrix that does not factorise into site charges (no potential field)"
                             : (getenv("SMOLMC_FORCE_GENERAL") || getenv("SMOLMC_NO_LEAN_MULTI")) ? "environment override" : "";
        if (!lean && h->lean_tables && (!h->lean_kf || wl) && multi_wl_ok && multi_bias_ok && Fk <= 64 && t->n_sublattices <= 4 && (!t->has_ewald || kp.ew_field) &&
            getenv("SMOLMC_FORCE_GENERAL") == nullptr && getenv("SMOLMC_NO_LEAN_MULTI") == nullptr) {
            LeanParams &lp = h->lp;
            const int ns = t->n_sublattices;
            bool ok = true;
            std::vector<double> mu_rows(32, 0.0), q_rows(32, 0.0), dg_rows(32, 0.0), bias_pairs((size_t)256 * SMOLMC_MAX_BIAS_ROWS, 0.0);
            double cum = 0.0, mmax = 0.0;
            for (int k = 0; k < ns && ok; ++k) {
                const int64_t a0 = t->sub_site_ptr[k], a1 = t->sub_site_ptr[k + 1];
                const int na = (int)(a1 - a0);
                int ncod = (int)(t->sub_code_ptr[k + 1] - t->sub_code_ptr[k]);
                const int sb = t->sub_active_sites[a0];
                auto no = [&](const char *why) { if (ok) h->lean_reason = why; ok = false; };
                if (na <= 0 || ncod < 2 || ncod > 8) no("an active sublattice of fewer than 2 or more than 8 species");
                for (int i = 0; ok && i < na; ++i)
                    if (t->sub_active_sites[a0 + i] != sb + i) no("the active sites of a sublattice are not one site range (Ensemble.make_tables(contiguous=True) relabels them)");
                {   // (codes other than 0 .. n-1 -- a sublattice split by species -- under canonical swaps only: see the single-class path)
                    bool default_codes = true;
                    int top = 0;
                    for (int c = 0; c < ncod; ++c) {
                        default_codes = default_codes && t->sub_codes[t->sub_code_ptr[k] + c] == c;
                        top = std::max(top, (int)t->sub_codes[t->sub_code_ptr[k] + c]);
                    }
                    if (ok && !default_codes) {
                        if (cfg->step_type == SMOLMC_STEP_SWAP && top < 8 && ncod >= 2 && getenv("SMOLMC_NO_SWAP_ANY_CODES") == nullptr) ncod = top + 1;
                        else no("a sublattice whose species codes are not 0 .. n-1 (split by species) under a step type that draws codes");
                    }
                }
                const int cls = ok ? h->site_class_host[sb] : 255;
                if (ok && cls == 255) no("an active sublattice without clusters");
                for (int i = 0; ok && i < na; ++i)
                    if (h->site_class_host[sb + i] != cls) no("sites of one sublattice in different site classes (cluster environments)"); // classes == sublattices
                if (!ok) break;
                lp.m_sbase[k] = sb; lp.m_nact[k] = na; lp.m_ncodes[k] = ncod; lp.m_cls[k] = cls;
                cum += t->sub_probs[k];
                lp.m_cum[k] = cum;
                if (t->has_mu) {
                    if (t->mu_width < ncod) ok = false;
                    for (int c = 0; ok && c < ncod; ++c) {
                        const double v = t->mu_table[(size_t)sb * t->mu_width + c];
                        mu_rows[k * 8 + c] = v;
                        mmax = std::max(mmax, std::fabs(v));
                        for (int i = 0; i < na; ++i)
                            if (t->mu_table[(size_t)(sb + i) * t->mu_width + c] != v) ok = false;
                    }
                }
                if (t->bias_type) { // one bias row per sublattice (it is defined per sublattice)
                    const int W = t->bias_width;
                    if (W < ncod) ok = false;
                    for (int rw = 0; rw < brows_l; ++rw) { // (rows of a hyperplane bias: [row][sublattice][old * 8 + new])
                        const double *tabk = h->bias_host.data() + (size_t)rw * t->num_sites * W;
                        for (int i = 0; ok && i < na; ++i)
                            for (int c = 0; c < ncod; ++c)
                                if (tabk[(size_t)(sb + i) * W + c] != tabk[(size_t)sb * W + c]) ok = false;
                        for (int o = 0; ok && o < ncod; ++o)
                            for (int n = 0; n < ncod; ++n) {
                                const double a = tabk[(size_t)sb * W + n], b = tabk[(size_t)sb * W + o];
                                bias_pairs[(size_t)rw * 256 + k * 64 + o * 8 + n] = t->bias_type == SMOLMC_BIAS_FUGACITY ? std::log(a / b) : a - b;
                            }
                    }
                }
                if (t->has_ewald) {
                    if (kp.ew_W > 8 || sb < kp.ew_act_base || sb + na > kp.ew_act_base + kp.ew_nact) ok = false;
                    for (int c = 0; ok && c < kp.ew_W; ++c) {
                        q_rows[k * 8 + c] = h->ew_qs_host[(size_t)sb * kp.ew_W + c];
                        dg_rows[k * 8 + c] = h->ew_dg_host[(size_t)sb * kp.ew_W + c];
                        for (int i = 0; i < na; ++i)
                            if (h->ew_qs_host[(size_t)(sb + i) * kp.ew_W + c] != q_rows[k * 8 + c] ||
                                h->ew_dg_host[(size_t)(sb + i) * kp.ew_W + c] != dg_rows[k * 8 + c])
                                ok = false;
                    }
                }
            }
            int ndims = 0, max_nact = 0;
            for (int k = 0; k < ns && ok; ++k) { ndims += lp.m_ncodes[k]; max_nact = std::max(max_nact, lp.m_nact[k]); }
            if (table && ok) {
                if (t->n_flip_vectors <= 0 || !t->flip_table || !t->flip_weights) ok = false;
                else if (t->n_flip_vectors > 8 || ndims > 16) ok = false;
            }
            // LDS: shared tables + slot records, per wave occupancy + scratch + accumulators (+ field)
            const size_t nrec = (size_t)h->lean_ncls * h->lean_nslot * 64;
            // (Wang-Landau: + feature scale and feature index of every slot record)
            const size_t shared = ((size_t)lp.dt_len + 96 + (table ? 80 : 0)) * 8 + nrec * 24 + (wl ? nrec * 12 : 0);
            // waves per workgroup: the shared tables are paid once per workgroup, so pick the size
            // that keeps the most waves resident per CU (160 KiB of LDS)
            // ... and, at equal residency, the size that spreads the walkers over the most CUs: 1024 walkers in
            // eight-wave workgroups are 128 workgroups -- half the chip idle, two waves per SIMD on the other half
            int cus_l = 0;
            if (hipDeviceGetAttribute(&cus_l, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus_l = 256;
            auto layout = [&](size_t per_wave_bytes, int &wpb_out) {
                int best_waves = 0;
                long best_rounds = 0, best_busy = 0;
                wpb_out = 0;
                for (int w : {8, 4, 2, 1}) {
                    const size_t need = shared + per_wave_bytes * w;
                    if (need > 160 * 1024 - 512) continue;
                    const int waves = (int)((160 * 1024) / need) * w;
                    const long R_ = cfg->n_replicas, blocks = (R_ + w - 1) / w;
                    const long rounds = (R_ + (long)waves * cus_l - 1) / ((long)waves * cus_l);
                    const long busy = std::min<long>(blocks, cus_l); // CUs that get a workgroup in the first round
                    const bool better = wpb_out == 0 || rounds < best_rounds ||
                                        (rounds == best_rounds && (busy > best_busy || (busy == best_busy && waves > best_waves)));
                    if (better) { best_waves = waves; best_rounds = rounds; best_busy = busy; wpb_out = w; }
                }
                return best_waves;
            };
            // The potential field goes to LDS while it stays small beside the rest of the wave's
            // state, when all walkers still fit the chip in one round with it there (an accepted
            // flip then reads its G row only, no read-modify-write of phi through L2), or for
            // canonical swaps whenever it fits; else the HBM copy is used in place (ew_field 2).  SMOLMC_MULTI_PHI_HBM / _LDS force either (test hooks).
            // (Wang-Landau: entropies, step counts and cached rows instead of the accumulator cells)
            const size_t base_wave = (size_t)lp.Nlds + 64 + 64 * 8 +
                                     (wl ? wl_multi_wave_bytes(h->L, h->F, wl_sum_mode) + (table ? nrec * 8 : 0) // (table: + pending cells)
                                         : nrec * (table ? 16 : 8));
            bool phi_lds = t->has_ewald && (size_t)kp.ew_nact * 8 <= base_wave / 2;
            if (t->has_ewald && !phi_lds) {
                int cus = 0, w = 0;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 256;
                const int waves = layout(base_wave + (size_t)kp.ew_nact * 8, w);
                // canonical swaps update the field at two sites per accepted step and gain from
                // the LDS copy even when the walkers then need several rounds (LiNiO2 8^3, 4096
                // walkers: 1.13e9 -> 1.58e9 steps/s); flips only while one round holds them all
                // (beyond: 4.1e9 with the HBM field against 3.0e9)
                // -- provided at least 8 waves per CU stay resident: at 3 (config 7, 27 KiB of field
                // per walker) the LDS copy measured 9.6 against 7.4 us per step
                // (Wang-Landau accepts three steps of four: the field update is the step, the LDS copy pays for flips too)
                if ((long)waves * cus >= (long)cfg->n_replicas || ((cfg->step_type == SMOLMC_STEP_SWAP || wl) && waves >= 8) ||
                    getenv("SMOLMC_MULTI_PHI_LDS") != nullptr)
                    phi_lds = waves > 0;
            }
            if (getenv("SMOLMC_MULTI_PHI_HBM") != nullptr) phi_lds = false;
            const size_t per_wave = base_wave + (phi_lds ? (size_t)kp.ew_nact * 8 : 0);
            int wpb = 0;
            layout(per_wave, wpb);
            if (wpb == 0 && ok) { ok = false; h->lean_reason = "walker state beyond the workgroup's LDS (multi-class layout)"; }
            if (!ok && h->lean_reason.empty())
                h->lean_reason = "sublattice layout (an active sublattice is not one site range of one class with codes 0..n-1, or per-site "
                                 "chemical potentials / bias / charges differ inside a sublattice), or a flip table beyond 8 vectors / 16 dims";
            if (ok) {
                if (t->has_mu && dev_upload(h, mu_rows.data(), 32, &lp.m_mu)) return bail(1);
                if (t->has_ewald &&
                    (dev_upload(h, q_rows.data(), 32, &lp.m_q) || dev_upload(h, dg_rows.data(), 32, &lp.m_dg)))
                    return bail(1);
                if (t->bias_type) {
                    if (dev_upload(h, bias_pairs.data(), bias_pairs.size(), &lp.bias_pair)) return bail(1);
                    lp.bias_rows = brows_l;
                    lp.bias_row_stride = 256;
                    lp.bias_type = t->bias_type;
                    lp.bias_pen = t->bias_penalty;
                    lp.bias = kp.bias;
                    lp.charge = kp.charge;
                }
                if (t->has_mu) lp.fast_eps += 4.0 * mmax * ldexp(1.0, -19);
                if (getenv("SMOLMC_NO_FAST_ACCEPT")) lp.fast_eps = 0.0;
                if (const char *sc = getenv("SMOLMC_FAST_EPS_SCALE")) lp.fast_eps *= atof(sc);
                lp.m_ncls = h->lean_ncls; lp.m_nsub = ns; lp.m_ndims = ndims;
                if (table) {
                    std::vector<double> ln((size_t)max_nact + 1, 0.0);
                    for (int k = 1; k <= max_nact; ++k) ln[k] = std::log((double)k);
                    if (dev_upload(h, t->flip_table, (size_t)t->n_flip_vectors * ndims, &lp.tf_table) ||
                        dev_upload(h, t->flip_weights, (size_t)2 * t->n_flip_vectors, &lp.tf_w) ||
                        dev_upload(h, ln.data(), ln.size(), &lp.tf_ln))
                        return bail(1);
                    lp.tf_n = t->n_flip_vectors;
                    lp.tf_sw = t->swap_weight;
                    lp.tf_ln_len = 0;
                }
                lp.occ = kp.occ; lp.enthalpy = kp.enthalpy; lp.features = h->lazy_tables ? h->d_lazy_scal : kp.features; lp.beta = kp.beta;
                lp.seeds = kp.seeds; lp.nsteps = kp.nsteps; lp.nacc = kp.nacc; lp.last_acc = kp.last_acc;
                lp.R = h->R; lp.N = h->N; lp.Npad = h->Npad; lp.F = Fk; lp.Fce = h->lazy_tables ? 0 : h->Fce;
                if (wl) {
                    lp.wl.L = h->L;
                    lp.wl.vmin = kp.wl_min; lp.wl.vmax = kp.wl_max; lp.wl.bin = kp.wl_bin;
                    lp.wl.flat = kp.wl_flat; lp.wl.div = kp.wl_div;
                    lp.wl.check = kp.wl_check; lp.wl.update = kp.wl_update;
                    lp.wl.entropy = kp.wl_entropy; lp.wl.hist = kp.wl_hist; lp.wl.occur = kp.wl_occur;
                    lp.wl.meanf = kp.wl_meanf; lp.wl.m = kp.wl_m; lp.wl.counter = kp.wl_counter;
                    lp.wl.sum_mode = wl_sum_mode;
                    h->lean_multi_wl = true;
                }
                lp.ew_field = 0;
                if (t->has_ewald) {
                    lp.ew_W = kp.ew_W; lp.ew_nact = kp.ew_nact; lp.ew_act_base = kp.ew_act_base;
                    lp.ew_G = kp.ew_G; lp.ew_coef = kp.ew_coef; lp.ew_phi = kp.ew_phi;
                    lp.ew_gx = kp.ew_gx; lp.ew_E8 = kp.ew_E8; lp.ew_S8 = kp.ew_S8;
                    lp.ew_field = phi_lds ? 1 : 2;
                    lp.sbase = kp.ew_act_base; // field_apply indexes phi relative to it
                }
                h->lean_lds = shared + per_wave * wpb;
                h->waves_per_block_lean = wpb;
                h->lean = true;
                h->lean_multi = true;
            }
        }
        if (getenv("SMOLMC_DEBUG"))
            fprintf(stderr, "[smolmc] lean=%d tables=%d nslot=%d mm=%d lds=%zu ew=%d compact=%d field=%d nact=%d "
                            "ew_nact=%d ew_act_base=%d sbase=%d general: nslot=%d mm=%d lds=%zu\n",
                    (int)lean, (int)h->lean_tables, h->lean_nslot, h->lean_mm, h->lean_lds, t->has_ewald,
                    kp.ew_compact, h->lp.ew_field, nact, kp.ew_nact, kp.ew_act_base, sbase, h->nslot, h->mm,
                    h->lds_bytes);
    }
    // ---- universal kernel (mc_univ.h): parameter block of every handle; `univ` when nothing else serves
    {
        UParams &up = h->up;
        memset(&up, 0, sizeof(up));
        up.T = h->rt;
        up.natural = h->d_natural;
        up.wl = wl ? 1 : 0;
        {
            const bool cm = t->feature_mode == SMOLMC_FEATURES_CORRELATIONS;
            const int64_t nloc = t->site_ptr[t->num_sites];
            std::vector<URec> recs((size_t)std::max<int64_t>(nloc, 1));
            memset(recs.data(), 0, recs.size() * sizeof(URec));
            for (int64_t r = 0; r < nloc; ++r) {
                const int o = t->loc_orbit[r];
                URec &u = recs[(size_t)r];
                u.I = t->orb_nsites[o];
                u.K = cm ? t->orb_nfunc[o] : 1;
                u.Nt = t->orb_tensor_len[o];
                u.J = t->loc_nrows[r];
                for (int m = 0; m < u.I; ++m) u.st[m] = t->tensor_indices[t->orb_stride_off[o] + m];
                u.feat = cm ? t->orb_bit_id[o] : t->orb_id[o];
                u.idx_off = t->loc_off[r];
                u.t_off = cm ? t->orb_ctensor_off[o] : t->orb_itensor_off[o];
                u.scale = (double)t->size / t->loc_ratio[r] / (double)t->loc_nrows[r];
                u.ratio = t->loc_ratio[r];
            }
            if (dev_upload(h, recs.data(), recs.size(), &up.recs)) return bail(1);
            // the cluster rows of every site, packed for the enthalpy pass (URow / URecE); equal records once
            std::vector<URecE> rec_e;
            std::vector<int32_t> rec_of((size_t)std::max<int64_t>(nloc, 1), 0);
            std::map<std::string, int32_t> rec_index;
            up.max_I = 1;
            up.all_k1 = 1;
            up.tens_len = 0;
            uint64_t nrows = 0;
            for (int64_t r = 0; r < nloc; ++r) {
                const URec &u = recs[(size_t)r];
                URecE e;
                memset(&e, 0, sizeof(e));
                for (int m = 0; m < u.I; ++m) e.st[m] = u.st[m];
                e.K = u.K; e.Nt = u.Nt; e.feat = u.feat; e.scale = u.scale;
                e.nat0 = (u.feat >= 0 && u.feat < h->F) ? h->natural[(size_t)u.feat] : 0.0;
                // (offsets inside one tensor, k * Nt + index, stay below the total checked here)
                if ((uint64_t)u.t_off + (uint64_t)u.K * (uint64_t)u.Nt > 0xffffffffull)
                    return bail(fail("cluster tensors of more than 2^32 entries"));
                e.t_off = (uint32_t)u.t_off;
                up.tens_len = (int)std::min<uint64_t>(0x7fffffffull, std::max<uint64_t>((uint64_t)up.tens_len, (uint64_t)u.t_off + (uint64_t)u.K * (uint64_t)u.Nt));
                up.max_I = std::max(up.max_I, (int)u.I);
                if (u.K != 1) up.all_k1 = 0;
                nrows += (uint64_t)u.J;
                const std::string key((const char *)&e, sizeof(e));
                auto it = rec_index.find(key);
                if (it == rec_index.end()) {
                    it = rec_index.emplace(key, (int32_t)rec_e.size()).first;
                    rec_e.push_back(e);
                }
                rec_of[(size_t)r] = it->second;
            }
            if (rec_e.empty()) { URecE e; memset(&e, 0, sizeof(e)); rec_e.push_back(e); }
            up.n_recs_e = (int)rec_e.size();
            up.dict_lds = up.n_recs_e <= SMOLMC_UNIV_DICT_RECS && up.tens_len <= SMOLMC_UNIV_DICT_TENS && h->F <= SMOLMC_UNIV_DICT_NAT &&
                          getenv("SMOLMC_UNIV_NO_DICT") == nullptr;
            if (nloc > 0x7fffffffll || nrows > 0x7fffffffull) return bail(fail("more than 2^31 local cluster rows"));
            std::vector<uint32_t> row_ptr((size_t)t->num_sites + 1, 0);
            std::vector<URow> rows((size_t)std::max<uint64_t>(nrows, 1));
            memset(rows.data(), 0, rows.size() * sizeof(URow));
            size_t q = 0;
            for (int s = 0; s < t->num_sites; ++s) {
                for (int64_t r = t->site_ptr[s]; r < t->site_ptr[s + 1]; ++r) {
                    const int I = t->orb_nsites[t->loc_orbit[r]];
                    for (int j = 0; j < t->loc_nrows[r]; ++j) {
                        URow &w = rows[q++];
                        const int32_t *m = t->loc_idx + t->loc_off[r] + (int64_t)j * I;
                        for (int i = 0; i < SMOLMC_MAX_CLUSTER_SITES; ++i) w.x[i] = i < I ? m[i] : (I ? m[0] : s);
                        w.rec = rec_of[(size_t)r];
                    }
                }
                row_ptr[(size_t)s + 1] = (uint32_t)q;
            }
            up.rows_uniform = t->num_sites > 0 ? (int)row_ptr[1] : 0;
            for (int s = 0; s < t->num_sites; ++s)
                if (row_ptr[(size_t)s + 1] - row_ptr[(size_t)s] != (uint32_t)up.rows_uniform) up.rows_uniform = 0;
            const URow *d_rows = nullptr;
            const URow16 *d_rows16 = nullptr;
            // (experiment build + SMOLMC_UNIV_ROWS16=1: measured on config 2 forced onto this kernel, the 16-byte rows cut the
            // fetched bytes 25.0 -> 8.7 GB per launch of 4096 x 2000 swap steps and COST 4 % of the rate -- 11 % on config
            // 8 -- because the kernel is VALU-bound and the unpacking is six more vector instructions per row)
#ifdef SMOLMC_UNIV_ROWS16
            up.rows16 = (t->num_sites <= 65535 && getenv("SMOLMC_UNIV_ROWS16") != nullptr) ? 1 : 0;
#else
            up.rows16 = 0; // (the kernels of the default build do not read 16-byte rows: make EXTRA=-DSMOLMC_UNIV_ROWS16)
#endif
            if (up.rows16) {
                std::vector<URow16> r16(rows.size());
                for (size_t i = 0; i < rows.size(); ++i) {
                    for (int k = 0; k < SMOLMC_MAX_CLUSTER_SITES; ++k) r16[i].x[k] = (uint16_t)rows[i].x[k];
                    r16[i].rec_lo = (uint16_t)((uint32_t)rows[i].rec & 0xffffu);
                    r16[i].rec_hi = (uint16_t)((uint32_t)rows[i].rec >> 16);
                }
                if (dev_upload(h, r16.data(), r16.size(), &d_rows16)) return bail(1);
            } else if (dev_upload(h, rows.data(), rows.size(), &d_rows)) return bail(1);
            if (dev_upload(h, row_ptr.data(), row_ptr.size(), &up.row_ptr) ||
                dev_upload(h, rec_e.data(), rec_e.size(), &up.recs_e))
                return bail(1);
            up.rows = up.rows16 ? (const uint4 *)d_rows16 : (const uint4 *)d_rows;
        }
        if (cfg->step_type == SMOLMC_STEP_TABLE_FLIP) {
            if (t->n_flip_vectors <= 0 || !t->flip_table || !t->flip_weights)
                return bail(fail("TableFlip needs a flip table (CompositionSpace.flip_table, "
                                 "smol/moca/composition/space.py:404-429)"));
            if (!(t->swap_weight >= 0.0 && t->swap_weight < 1.0))
                return bail(fail("swap_weight must be in [0, 1)"));
            const int ns = t->n_sublattices, d = (int)t->sub_code_ptr[ns];
            if (t->n_flip_vectors > SMOLMC_MAX_FLIP_VECTORS || d > SMOLMC_MAX_FLIP_DIMS)
                return bail(fail("flip table larger than SMOLMC_MAX_FLIP_VECTORS x SMOLMC_MAX_FLIP_DIMS"));
            std::vector<int> dim_sub(d);
            int max_nact = 0;
            for (int k = 0; k < ns; ++k) {
                for (int64_t c = t->sub_code_ptr[k]; c < t->sub_code_ptr[k + 1]; ++c) dim_sub[c] = k;
                max_nact = std::max(max_nact, (int)(t->sub_site_ptr[k + 1] - t->sub_site_ptr[k]));
            }
            h->max_step_flips = 2;
            for (int v = 0; v < t->n_flip_vectors; ++v) {
                // species are exchanged inside a sublattice (mcusher.py:612-634: "Sub-lattices can not
                // cross", assert len(site_ids) == 0), a step flips the sites of the depleted species
                int total = 0;
                for (int k = 0; k < ns; ++k) {
                    int sum = 0, picks = 0;
                    for (int64_t c = t->sub_code_ptr[k]; c < t->sub_code_ptr[k + 1]; ++c) {
                        const int u = t->flip_table[(size_t)v * d + c];
                        sum += u;
                        picks += u < 0 ? -u : 0;
                    }
                    if (sum != 0) return bail(fail("flip vector does not conserve the sites of a sublattice"));
                    total += picks;
                }
                if (total == 0) return bail(fail("flip vector without any flip"));
                if (total > SMOLMC_MAX_STEP_FLIPS) return bail(fail("flip vector of more than SMOLMC_MAX_STEP_FLIPS flips"));
                h->max_step_flips = std::max(h->max_step_flips, total);
            }
            for (int i = 0; i < 2 * t->n_flip_vectors; ++i)
                if (!(t->flip_weights[i] >= 0.0)) return bail(fail("flip weights must be non-negative"));
            std::vector<double> ln((size_t)max_nact + 2, 0.0);
            for (int k = 1; k <= max_nact + 1; ++k) ln[k] = std::log((double)k);
            if (dev_upload(h, t->flip_table, (size_t)t->n_flip_vectors * d, &up.tf_table) ||
                dev_upload(h, t->flip_weights, (size_t)2 * t->n_flip_vectors, &up.tf_w) ||
                dev_upload(h, ln.data(), ln.size(), &up.tf_ln) || dev_upload(h, dim_sub.data(), dim_sub.size(), &up.tf_dim_sub))
                return bail(1);
            up.tf_n = t->n_flip_vectors;
            up.tf_d = d;
            up.tf_sw = t->swap_weight;
        }
        // per-wave LDS: step scratch (flips, counts, weights, a-priori factors, running sums: 2464 B) + the occupancy when it fits
        // (+ the step's feature deltas, one cell per cluster feature, while that stays small)
        up.dfeat_cells = (h->Fce <= 1024 && getenv("SMOLMC_UNIV_TWO_PASS") == nullptr) ? (h->Fce + 1) / 2 * 2 : 0;
        up.acc_cells = up.dfeat_cells ? (h->F + 1) / 2 * 2 : 0;
        up.lds_shared = up.dict_lds ? SMOLMC_UNIV_DICT_BYTES : 0;
        auto lay_out = [&]() {
            const size_t scratch = (cfg->step_type == SMOLMC_STEP_TABLE_FLIP ? SMOLMC_UNIV_SCRATCH_TABLE : SMOLMC_UNIV_SCRATCH) +
                                   (((size_t)up.dfeat_cells << up.dfeat_shift) + (size_t)up.acc_cells) * 8,
                         with_occ = (scratch + (size_t)h->Npad + 15) & ~(size_t)15;
            up.occ_lds = with_occ <= 160 * 1024 - 256 && getenv("SMOLMC_UNIV_OCC_HBM") == nullptr;
            up.lds_per_wave = (int)(up.occ_lds ? with_occ : scratch);
            h->univ_wpb = 4;
            while (h->univ_wpb > 1 && (size_t)up.lds_shared + (size_t)up.lds_per_wave * h->univ_wpb > 64 * 1024) h->univ_wpb /= 2;
        };
        // shadow copies of the step's feature cells: as many as keep them within 4 KB -- and, for launches of more than
        // twelve walkers per CU, the workgroup within a quarter of the CU's LDS (a workgroup of 40.1 KB halves the resident
        // walkers of config 2: 1.3e9 -> 8.1e8 flips/s)
        up.dfeat_shift = 0;
        while (up.dfeat_cells && getenv("SMOLMC_UNIV_NO_COPIES") == nullptr && up.dfeat_shift < 3 && ((size_t)up.dfeat_cells << (up.dfeat_shift + 1)) <= 512) up.dfeat_shift++;
        lay_out();
        const bool crowded = (long long)cfg->n_replicas > 12ll * 256;
        auto too_big = [&]() { return crowded && h->univ_wpb == 4 && (size_t)up.lds_shared + (size_t)up.lds_per_wave * 4 > 40 * 1024; };
        while (too_big() && up.dfeat_shift > 0) { up.dfeat_shift--; lay_out(); }
        if (too_big() && up.dict_lds) {
            up.dict_lds = 0;
            up.lds_shared = 0;
            lay_out();
        }
        const bool table = cfg->step_type == SMOLMC_STEP_TABLE_FLIP;
        h->univ = !h->lean && (table || !h->general_ok);
    }
    *out = h;
    return 0;
}

extern "C" int smolmc_destroy(smolmc_handle *h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) hipFree(p);
    if (h->d_eval_occ) hipFree(h->d_eval_occ);
    free_samples(h);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

extern "C" int smolmc_num_features(const smolmc_handle *h) { return h ? h->F : -1; }
extern "C" int smolmc_wl_num_levels(const smolmc_handle *h) { return h ? h->L : -1; }
extern "C" int smolmc_natural_parameters(const smolmc_handle *h, double *out) {
    if (!h || !out) return fail("null argument");
    memcpy(out, h->natural.data(), h->natural.size() * 8);
    return 0;
}

extern "C" int smolmc_set_stream(smolmc_handle *h, void *stream) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream) hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return 0;
}

static int launch_eval_full(smolmc_handle *h, const uint8_t *d_occ8, int nocc, double *d_out, int ce_only = 0) {
    // occupancies per block (see the kernel): four when the copies fit 60 KB of LDS and only the cluster features are
    // wanted, one with the scalar features (their loops take one occupancy), none staged beyond 60 KB
    int M = h->Npad <= 60 * 1024 ? 1 : 0;
    if (ce_only && M && nocc >= 8) M = std::min(4, (60 * 1024) / h->Npad);
    const int per = M > 0 ? M : 1;
    hipLaunchKernelGGL(eval_full_kernel, dim3((unsigned)((nocc + per - 1) / per)), dim3(256), (size_t)M * h->Npad, h->stream, h->rt, d_occ8,
                       d_out, ce_only, nocc, M);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- lazy cluster features (build_mc_tables) ---------------------------------------------------------------
// The lean kernels of such a handle carry the scalar features only (d_lazy_scal, row stride nscal); the cluster
// part of kp.features is evaluated from the occupancies where it is read.
static bool is_lazy(const smolmc_handle *h) { return h->lazy_tables && h->lean; }
static int lazy_nscal(const smolmc_handle *h) { return (h->rt.has_ewald ? 1 : 0) + (h->rt.has_mu ? 1 : 0); }
// rows of full feature vectors [n][F] <-> rows of scalar features [n][nscal]
__global__ void lazy_scalars_kernel(double *features, double *scal, size_t n, int F, int Fce, int nscal, int to_scal) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (size_t)nscal) return;
    const size_t row = i / nscal;
    const int j = (int)(i % nscal);
    if (to_scal) scal[row * nscal + j] = features[row * F + Fce + j];
    else features[row * F + Fce + j] = scal[row * nscal + j];
}
static int lazy_scalars(smolmc_handle *h, double *features, double *scal, size_t n, int to_scal) {
    const int ns = lazy_nscal(h);
    if (!ns || !n) return 0;
    hipLaunchKernelGGL(lazy_scalars_kernel, dim3((unsigned)((n * ns + 255) / 256)), dim3(256), 0, h->stream, features, scal, n,
                       h->F, h->Fce, ns, to_scal);
    HIPCHK(hipGetLastError());
    return 0;
}
// kp.features current (before anybody reads them, and before a kernel that updates them incrementally -- mc_kernel /
// the universal kernel replaying records the lean kernels do not take -- runs on a lazy handle)
static int ensure_features(smolmc_handle *h) {
    if (!is_lazy(h) || !h->ce_dirty) return 0;
    TRY(launch_eval_full(h, h->kp.occ, h->R, h->kp.features, 1));
    TRY(lazy_scalars(h, h->kp.features, h->d_lazy_scal, (size_t)h->R, 0));
    h->ce_dirty = false;
    return 0;
}
// ... and after such a kernel (or smolmc_set_state): the scalars the lean kernels continue from
static int lazy_scalars_from_features(smolmc_handle *h) {
    if (!is_lazy(h)) return 0;
    TRY(lazy_scalars(h, h->kp.features, h->d_lazy_scal, (size_t)h->R, 1));
    h->ce_dirty = false;
    return 0;
}

// ---- the boundary of a relabelled handle (see plan_relabelling): caller's numbering <-> engine's ----------------
// rows of occupancies in the caller's numbering -> the engine's (a copy; the caller's array on a plain handle)
static const int32_t *occ_in(const smolmc_handle *h, const int32_t *occ, size_t rows, std::vector<int32_t> &tmp) {
    if (!h->relabelled) return occ;
    const size_t N = (size_t)h->N;
    tmp.resize(rows * N);
    const int32_t *old_of = h->old_of.data();
    for (size_t r = 0; r < rows; ++r) {
        const int32_t *src = occ + r * N;
        int32_t *dst = tmp.data() + r * N;
        for (size_t p = 0; p < N; ++p) dst[p] = src[old_of[p]];
    }
    return tmp.data();
}
// ... and back, in place
static void occ_out(const smolmc_handle *h, int32_t *occ, size_t rows) {
    if (!h->relabelled) return;
    const size_t N = (size_t)h->N;
    std::vector<int32_t> row(N);
    const int32_t *new_of = h->new_of.data();
    for (size_t r = 0; r < rows; ++r) {
        int32_t *o = occ + r * N;
        std::copy(o, o + N, row.begin());
        for (size_t p = 0; p < N; ++p) o[p] = row[new_of[p]];
    }
}
// step records (site, code) x SMOLMC_MAX_STEP_FLIPS: the sites renamed; a site outside the cell is left as it is --
// the range checks of the entry point reject it (renaming it would turn it into a step nobody asked for)
static const int32_t *steps_in(const smolmc_handle *h, const int32_t *steps, size_t nrec, std::vector<int32_t> &tmp) {
    if (!h->relabelled) return steps;
    tmp.assign(steps, steps + nrec * SMOLMC_STEP_ROW);
    for (size_t i = 0; i < nrec; ++i)
        for (int f = 0; f < SMOLMC_MAX_STEP_FLIPS; ++f) {
            int32_t &site = tmp[i * SMOLMC_STEP_ROW + 2 * f];
            if (site < 0) break;
            if (site < h->N) site = h->new_of[site];
        }
    return tmp.data();
}

static int upload_occ(smolmc_handle *h, const int32_t *occ, size_t nocc, uint8_t *d_occ8) {
    const size_t n32 = nocc * h->N;
    // every code must exist on its site: a larger one would index past the tensors / tables on the
    // device (the reference reads garbage there, SURVEY 8b "error conventions")
    const uint8_t *lim = h->site_ncodes.data();
    const size_t N = (size_t)h->N;
    for (size_t i = 0, s = 0; i < n32; ++i, s = (s + 1 == N ? 0 : s + 1))
        if (occ[i] < 0 || occ[i] >= (int)lim[s]) {
            char msg[160];
            snprintf(msg, sizeof msg, "occupancy code %d out of range on site %zu (%d species codes there)", occ[i],
                     h->relabelled ? (size_t)h->old_of[s] : s, (int)lim[s]); // (the site in the caller's numbering)
            return fail(msg);
        }
    int *d32 = nullptr;
    HIPCHK(hipMalloc((void **)&d32, std::max<size_t>(n32 * 4, 16)));
    hipError_t e = hipMemcpyAsync(d32, occ, n32 * 4, hipMemcpyHostToDevice, h->stream);
    const size_t total = nocc * h->Npad;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pack_occ_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream,
                           d32, d_occ8, h->N, h->Npad, total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d32);
    if (e != hipSuccess) return fail(std::string("occupancy upload: ") + hipGetErrorString(e));
    return 0;
}

static void set_betas(smolmc_handle *h, const double *temperature, std::vector<double> &beta) {
    beta.resize(h->R);
    for (int r = 0; r < h->R; ++r) {
        const double T = temperature ? temperature[r] : 0.0;
        beta[r] = 1.0 / (SMOLMC_KB * T); // ThermalKernelMixin (kernel/base.py:398)
    }
}

extern "C" int smolmc_set_state(smolmc_handle *h, const int32_t *occ, const uint64_t *seeds,
                                const double *temperature, int reset_aux) {
    if (!h || !occ) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    KParams &kp = h->kp;
    const size_t R = h->R;
    std::vector<int32_t> occ_engine;
    occ = occ_in(h, occ, R, occ_engine);
    TRY(upload_occ(h, occ, R, kp.occ));
    std::vector<double> beta;
    set_betas(h, temperature, beta);
    HIPCHK(hipMemcpy(h->d_beta, beta.data(), R * 8, hipMemcpyHostToDevice));
    h->order_dirty = true;
    if (reset_aux) {
        std::vector<uint64_t> sd(R);
        for (size_t r = 0; r < R; ++r) sd[r] = seeds ? seeds[r] : (uint64_t)r;
        HIPCHK(hipMemcpy((void *)kp.seeds, sd.data(), R * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemsetAsync(kp.nsteps, 0, R * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.nacc, 0, R * 8, h->stream));
    }
    HIPCHK(hipMemsetAsync(kp.last_acc, 1, R, h->stream));
    if (kp.bias_type) {
        // MCBias.compute_bias of every initial occupancy (kernel/base.py:362-363), on the host:
        // the occupancies are host arrays here and this runs once per set_state
        std::vector<double> b0(R), q0(R * SMOLMC_MAX_BIAS_ROWS, 0.0);
        const int N = h->N, W = kp.bias_W;
        for (size_t r = 0; r < R; ++r) {
            const int32_t *o = occ + r * (size_t)N;
            if (kp.bias_type == SMOLMC_BIAS_FUGACITY) {
                double acc = 0.0;
                for (int s = 0; s < N; ++s) acc += log(h->bias_host[(size_t)s * W + o[s]]); // bias.py:174-186
                b0[r] = acc;
            } else {
                // SquareChargeBias (bias.py:264-277) = one hyperplane with intercept 0;
                // SquareHyperplaneBias (bias.py:352-366): -penalty * sum_r (A_r . n - b_r)^2
                double sq = 0.0;
                for (int k = 0; k < kp.bias_rows; ++k) {
                    double acc = 0.0;
                    const double *tab = h->bias_host.data() + (size_t)k * kp.bias_row_stride;
                    for (int s = 0; s < N; ++s) acc += tab[(size_t)s * W + o[s]];
                    acc -= h->bias_icpt[k];
                    q0[r * SMOLMC_MAX_BIAS_ROWS + k] = acc;
                    sq += acc * acc;
                }
                b0[r] = -kp.bias_pen * sq;
            }
        }
        HIPCHK(hipMemcpy(kp.bias, b0.data(), R * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(kp.charge, q0.data(), q0.size() * 8, hipMemcpyHostToDevice));
    }
    if (kp.ew_field) {
        LeanParams fp;
        memset(&fp, 0, sizeof(fp));
        fp.occ = kp.occ; fp.Npad = h->Npad; fp.sbase = kp.ew_act_base; fp.ew_nact = kp.ew_nact;
        fp.ew_W = kp.ew_W; fp.ew_qs = kp.ew_qs; fp.ew_G = kp.ew_G; fp.ew_frozen = kp.ew_frozen;
        fp.ew_phi = kp.ew_phi;
        const size_t shm = (size_t)kp.ew_nact * 8;
        if (shm > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)ewald_field_init_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(ewald_field_init_kernel, dim3((unsigned)R), dim3(256), shm, h->stream, fp);
        HIPCHK(hipGetLastError());
    }
    TRY(launch_eval_full(h, kp.occ, (int)R, kp.features));
    hipLaunchKernelGGL(dot_features_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, h->stream,
                       kp.features, h->d_natural, kp.enthalpy, (int)R, h->F);
    HIPCHK(hipGetLastError());
    TRY(lazy_scalars_from_features(h));
    if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU && reset_aux) {
        const size_t RL = R * h->L;
        HIPCHK(hipMemsetAsync(kp.wl_entropy, 0, RL * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.wl_hist, 0, RL * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.wl_occur, 0, RL * 8, h->stream));
        HIPCHK(hipMemsetAsync(kp.wl_meanf, 0, RL * h->F * 8, h->stream));
        h->wl_sums = false; // all zero: either representation
        HIPCHK(hipMemsetAsync(kp.wl_counter, 0, R * 8, h->stream));
        std::vector<double> m0(R, h->cfg.wl_mod_factor);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(kp.wl_m, m0.data(), R * 8, hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU) {
        // The bin of the CURRENT enthalpy indexes the entropy / histogram arrays unchecked in the
        // kernels (only proposed enthalpies are tested against the window, wanglandau.py:190-193).
        // The reference raises IndexError above the window and wraps to a wrong bin below it
        // (negative Python index, wanglandau.py:175-180); here both are refused up front.
        std::vector<double> H0(R);
        HIPCHK(hipMemcpy(H0.data(), kp.enthalpy, R * 8, hipMemcpyDeviceToHost));
        for (size_t r = 0; r < R; ++r)
            if (!(H0[r] >= h->cfg.wl_min_enthalpy && H0[r] < h->cfg.wl_max_enthalpy)) {
                char msg[200];
                snprintf(msg, sizeof msg,
                         "initial enthalpy %.6f of walker %zu is outside the Wang-Landau window "
                         "[min_enthalpy, max_enthalpy) = [%.6f, %.6f)",
                         H0[r], r, h->cfg.wl_min_enthalpy, h->cfg.wl_max_enthalpy);
                return fail(msg);
            }
    }
    return 0;
}

extern "C" int smolmc_set_temperature(smolmc_handle *h, const double *temperature) {
    if (!h || !temperature) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> beta;
    set_betas(h, temperature, beta);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d_beta, beta.data(), (size_t)h->R * 8, hipMemcpyHostToDevice));
    h->order_dirty = true;
    return 0;
}

extern "C" int smolmc_sync(smolmc_handle *h) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int smolmc_kernel_info(const smolmc_handle *h, char *buf, int n) {
    if (!h || !buf || n <= 0) return fail("null argument");
    if (h->univ)
        snprintf(buf, (size_t)n, "universal occ=%s field=%d lds=%zu (%s)", h->up.occ_lds ? "lds" : "hbm", h->kp.ew_field,
                 (size_t)h->up.lds_shared + (size_t)h->up.lds_per_wave * h->univ_wpb, h->general_ok ? "TableFlip outside the lean families" : h->general_reason.c_str());
    else if (h->lean)
        snprintf(buf, (size_t)n, "%s nslot=%d mm=%d field=%d lds=%zu%s", h->lean_multi ? "lean-multi" : "lean",
                 h->lean_nslot, h->lean_mm, h->lp.ew_field, h->lean_lds,
                 h->lean_solo ? (h->lean_occ ? " solo=1 occ=6" : " solo=1") : (h->lean_kf ? " kf=1" : ""));
    else
        snprintf(buf, (size_t)n, "general nslot=%d mm=%d field=%d lds=%zu", h->nslot, h->mm, h->kp.ew_field,
                 h->lds_bytes);
    {   // suffixes: the translation-compressed Ewald kernel in use, the Wang-Landau kernel generation
        const size_t used = strlen(buf);
        if (h->lean && h->lp.ew_field && h->lp.ew_gx && used + 40 < (size_t)n)
            snprintf(buf + used, (size_t)n - used, " gx=%dx%dx%dx%d", h->ew_gx_blocks, h->ew_gx_dims[0], h->ew_gx_dims[1],
                     h->ew_gx_dims[2]);
        else if (h->lean && h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU && used + 8 < (size_t)n)
            snprintf(buf + used, (size_t)n - used, (h->lean_multi_wl || h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP) ? "" : (getenv("SMOLMC_WL_V2") ? " wl=v2" : " wl=v3"));
        if (h->lean && !h->lean_multi && h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU && h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP &&
            strlen(buf) + 16 < (size_t)n) // (mc_table_kernel<..., WLT>)
            strncat(buf, h->lp.wl.sum_mode ? " wl=table" : " wl=table-mean", (size_t)n - strlen(buf) - 1);
        if (is_lazy(h) && strlen(buf) + 16 < (size_t)n) strncat(buf, " lazy-features", (size_t)n - strlen(buf) - 1);
        if (h->lean_multi_wl && strlen(buf) + 24 < (size_t)n) // (the Wang-Landau variant of the multi-class kernel)
            snprintf(buf + strlen(buf), (size_t)n - strlen(buf), h->lp.wl.sum_mode ? " wl=multi" : " wl=multi-mean");
        if (h->relabelled && strlen(buf) + 16 < (size_t)n) strncat(buf, " relabelled=1", (size_t)n - strlen(buf) - 1);
        // why the model runs neither lean family (the first condition that failed at smolmc_create)
        if (!h->lean && !h->lean_reason.empty() && strlen(buf) + h->lean_reason.size() + 16 < (size_t)n) {
            strncat(buf, " | not lean: ", (size_t)n - strlen(buf) - 1);
            strncat(buf, h->lean_reason.c_str(), (size_t)n - strlen(buf) - 1);
        }
    }
    return 0;
}

extern "C" int smolmc_get_bias(smolmc_handle *h, double *bias) {
    if (!h || !bias) return fail("null argument");
    if (!h->kp.bias_type) return fail("the model has no bias term");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(bias, h->kp.bias, (size_t)h->R * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int smolmc_get_state(smolmc_handle *h, int32_t *occ, double *features, double *enthalpy,
                                uint64_t *n_accepted, uint64_t *n_steps, uint8_t *last_accepted) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t R = h->R;
    KParams &kp = h->kp;
    if (occ) {
        int *d32 = nullptr;
        const size_t total = R * h->N;
        HIPCHK(hipMalloc((void **)&d32, total * 4));
        hipLaunchKernelGGL(unpack_occ_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           h->stream, kp.occ, d32, h->N, h->Npad, total);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(occ, d32, total * 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        hipFree(d32);
        if (e != hipSuccess) return fail(std::string("occupancy download: ") + hipGetErrorString(e));
        occ_out(h, occ, R);
    }
    if (features) {
        TRY(ensure_features(h));
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(features, kp.features, R * h->F * 8, hipMemcpyDeviceToHost));
    }
    if (enthalpy) HIPCHK(hipMemcpy(enthalpy, kp.enthalpy, R * 8, hipMemcpyDeviceToHost));
    if (n_accepted) HIPCHK(hipMemcpy(n_accepted, kp.nacc, R * 8, hipMemcpyDeviceToHost));
    if (n_steps) HIPCHK(hipMemcpy(n_steps, kp.nsteps, R * 8, hipMemcpyDeviceToHost));
    if (last_accepted) HIPCHK(hipMemcpy(last_accepted, kp.last_acc, R, hipMemcpyDeviceToHost));
    return 0;
}

// bring kp.wl_meanf into the representation the next consumer expects (see WlParams::sum_mode)
static int wl_set_representation(smolmc_handle *h, bool sums) {
    if (h->cfg.kernel_type != SMOLMC_KERNEL_WANGLANDAU || h->wl_sums == sums) return 0;
    const size_t cells = (size_t)h->R * h->L, n = cells * h->F;
    hipLaunchKernelGGL(wl_meanf_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream,
                       h->kp.wl_meanf, h->kp.wl_occur, cells, h->F, sums ? 0 : 1);
    HIPCHK(hipGetLastError());
    h->wl_sums = sums;
    return 0;
}

extern "C" int smolmc_get_wl(smolmc_handle *h, double *entropy, int64_t *histogram,
                             int64_t *occurrences, double *mean_features, double *mod_factor) {
    if (!h) return fail("null handle");
    if (h->cfg.kernel_type != SMOLMC_KERNEL_WANGLANDAU) return fail("handle is not a Wang-Landau kernel");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t RL = (size_t)h->R * h->L;
    KParams &kp = h->kp;
    TRY(wl_set_representation(h, false));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (entropy) HIPCHK(hipMemcpy(entropy, kp.wl_entropy, RL * 8, hipMemcpyDeviceToHost));
    if (histogram) HIPCHK(hipMemcpy(histogram, kp.wl_hist, RL * 8, hipMemcpyDeviceToHost));
    if (occurrences) HIPCHK(hipMemcpy(occurrences, kp.wl_occur, RL * 8, hipMemcpyDeviceToHost));
    if (mean_features)
        HIPCHK(hipMemcpy(mean_features, kp.wl_meanf, RL * h->F * 8, hipMemcpyDeviceToHost));
    if (mod_factor) HIPCHK(hipMemcpy(mod_factor, kp.wl_m, (size_t)h->R * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int smolmc_set_wl(smolmc_handle *h, const double *entropy, const int64_t *histogram,
                             const int64_t *occurrences, const double *mean_features, const double *mod_factor) {
    if (!h) return fail("null handle");
    if (h->cfg.kernel_type != SMOLMC_KERNEL_WANGLANDAU) return fail("handle is not a Wang-Landau kernel");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t RL = (size_t)h->R * h->L;
    KParams &kp = h->kp;
    // occurrences and mean features belong together: bring the device copy to running MEANS (what
    // the caller holds) before either is replaced
    TRY(wl_set_representation(h, false));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (mod_factor)
        for (int r = 0; r < h->R; ++r)
            if (!(mod_factor[r] > 0)) return fail("mod_factor must be greater than 0.");
    if (entropy) HIPCHK(hipMemcpy(kp.wl_entropy, entropy, RL * 8, hipMemcpyHostToDevice));
    if (histogram) HIPCHK(hipMemcpy(kp.wl_hist, histogram, RL * 8, hipMemcpyHostToDevice));
    if (occurrences) HIPCHK(hipMemcpy(kp.wl_occur, occurrences, RL * 8, hipMemcpyHostToDevice));
    if (mean_features) HIPCHK(hipMemcpy(kp.wl_meanf, mean_features, RL * h->F * 8, hipMemcpyHostToDevice));
    if (mod_factor) HIPCHK(hipMemcpy(kp.wl_m, mod_factor, (size_t)h->R * 8, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int smolmc_set_counters(smolmc_handle *h, const uint64_t *n_steps, const uint64_t *n_accepted) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    KParams &kp = h->kp;
    const size_t R = (size_t)h->R;
    if (n_steps) {
        HIPCHK(hipMemcpy(kp.nsteps, n_steps, R * 8, hipMemcpyHostToDevice));
        if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU) // the check period counts the same steps
            HIPCHK(hipMemcpy(kp.wl_counter, n_steps, R * 8, hipMemcpyHostToDevice));
    }
    if (n_accepted) HIPCHK(hipMemcpy(kp.nacc, n_accepted, R * 8, hipMemcpyHostToDevice));
    return 0;
}

// ---- kernel dispatch (instantiations live in general_n*.hip / lean_n*.hip) --------------
static int launch_mc(smolmc_handle *h, const KParams &kp, int replay) {
    switch (h->nslot) {
    case 2: return smolmc_launch_general_2(h, kp, replay);
    case 4: return smolmc_launch_general_4(h, kp, replay);
    case 8: return smolmc_launch_general_8(h, kp, replay);
    default: return smolmc_launch_general_16(h, kp, replay);
    }
}

// Launch-slot -> walker permutation for launches whose walkers differ in cost.  On an exchange ladder
// a hot walker accepts several times as many steps as a cold one and an accepted TableFlip step
// costs several rejected ones (field sweep, stale marking); a launch ends with its slowest SIMD,
// and a SIMD hosts one wave of each of the two workgroups resident on its CU.  Slots are therefore
// dealt so that the hottest walker shares its SIMD with the coldest, the second hottest with the
// second coldest, ...: every SIMD gets about the same work.  mode 1 (SMOLMC_WALKER_ORDER, default):
// the partner of a workgroup is the one half a grid later (workgroups are dealt round-robin over
// the CUs, the second half of the grid lands on the CUs of the first); mode 2: the next workgroup;
// 0: slot q runs walker q.  Ranks by beta ascending (hottest first), ties by walker index; built on
// the host whenever the temperatures changed (one 8 B-per-walker read-back per exchange sweep).
static int update_walker_order(smolmc_handle *h, LeanParams &lp) {
    const char *env = getenv("SMOLMC_WALKER_ORDER"); // (read at every launch: tests switch it)
    const int mode = env ? atoi(env) : 1;
    if (mode != h->order_mode) h->order_dirty = true;
    h->order_mode = mode;
    lp.order = nullptr;
    const int R = h->R, wpb = h->lean_wpb;
    if (mode == 0 || h->cfg.step_type != SMOLMC_STEP_TABLE_FLIP || h->lean_multi || R < 2 * wpb) return 0;
    if (!h->d_order) {
        HIPCHK(hipMalloc((void **)&h->d_order, (size_t)R * sizeof(int)));
        h->allocs.push_back(h->d_order);
        h->order_dirty = true;
    }
    if (h->order_dirty) {
        std::vector<double> beta(R);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(beta.data(), h->d_beta, (size_t)R * 8, hipMemcpyDeviceToHost));
        std::vector<int> by_rank(R), order(R);
        for (int i = 0; i < R; ++i) by_rank[i] = i;
        std::stable_sort(by_rank.begin(), by_rank.end(), [&](int a, int b) { return beta[a] < beta[b]; }); // hottest first
        const int nb = (R + wpb - 1) / wpb;
        for (int q = 0; q < R; ++q) {
            int rho;
            if (wpb == 8) {  // partner of wave w of a workgroup: wave w + 4 of the same workgroup (R % 8 == 0)
                const int k = (q >> 3) * 4 + (q & 3);
                rho = (q & 4) ? R - 1 - k : k;
            } else if (mode == 1) { // partner of workgroup b: workgroup b + nb / 2
                const int half = (nb / 2) * wpb;
                rho = q < half ? q : (q < 2 * half ? R - 1 - (q - half) : q - half);
            } else {         // partner of workgroup 2 p: workgroup 2 p + 1
                const int b = q / wpb, w = q % wpb, k = (b >> 1) * wpb + w;
                const bool paired = (b | 1) < nb;
                rho = !paired ? (nb / 2) * wpb + w : ((b & 1) ? R - 1 - k : k);
            }
            order[q] = by_rank[rho];
        }
        { // (a permutation by construction; a slip here would run a walker twice and another not at all)
            std::vector<char> seen(R, 0);
            for (int q = 0; q < R; ++q) {
                if (order[q] < 0 || order[q] >= R || seen[order[q]]) return fail("internal: walker order is not a permutation");
                seen[order[q]] = 1;
            }
        }
        HIPCHK(hipMemcpy(h->d_order, order.data(), (size_t)R * sizeof(int), hipMemcpyHostToDevice));
        h->order_dirty = false;
    }
    lp.order = h->d_order;
    return 0;
}

static int launch_lean(smolmc_handle *h, LeanParams lp, int64_t nsteps) {
    lp.steps = nsteps;
    TRY(update_walker_order(h, lp));
    if (h->lean_multi_wl && h->lean_kf) // several correlation functions per orbit (KFW)
        return h->lean_nslot == 2 ? smolmc_launch_multi_wl_kf_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_wl_kf_4(h, lp) : smolmc_launch_multi_wl_kf_8(h, lp));
    if (h->lean_multi_wl && h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP)
        return h->lean_nslot == 2 ? smolmc_launch_multi_table_wl_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_table_wl_4(h, lp) : smolmc_launch_multi_table_wl_8(h, lp));
    if (h->lean_multi_wl)
        return h->lean_nslot == 2 ? smolmc_launch_multi_wl_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_wl_4(h, lp) : smolmc_launch_multi_wl_8(h, lp));
    if (h->lean_multi && lp.bias_type && h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP)
        return h->lean_nslot == 2 ? smolmc_launch_multi_table_bias_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_table_bias_4(h, lp) : smolmc_launch_multi_table_bias_8(h, lp));
    if (h->lean_multi && lp.bias_type)
        return h->lean_nslot == 2 ? smolmc_launch_multi_bias_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_bias_4(h, lp) : smolmc_launch_multi_bias_8(h, lp));
    if (h->lean_multi)
        return h->lean_nslot == 2 ? smolmc_launch_multi_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_4(h, lp) : smolmc_launch_multi_8(h, lp));
    if (lp.bias_type && h->cfg.step_type != SMOLMC_STEP_TABLE_FLIP)
        return h->lean_nslot == 2 ? smolmc_launch_lean_bias_2(h, lp) : smolmc_launch_lean_bias_4(h, lp);
    if (lp.bias_type) // TableFlip with an MCBias term (single-class layout)
        return h->lean_nslot == 2 ? smolmc_launch_table_bias_2(h, lp) : smolmc_launch_table_bias_4(h, lp);
    if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU && h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP) // (single-class layout)
        return h->lean_nslot == 2 ? smolmc_launch_table_wl_2(h, lp) : smolmc_launch_table_wl_4(h, lp);
    if (h->lean_kf) return h->lean_nslot == 2 ? smolmc_launch_lean_corr_2(h, lp) : smolmc_launch_lean_corr_4(h, lp);
    // Wang-Landau: the dedicated kernel (mc_wl.h); SMOLMC_WL_V2 keeps round 2's variant of
    // mc_lean_kernel reachable for A/B runs
    if (h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU && h->cfg.step_type != SMOLMC_STEP_TABLE_FLIP &&
        getenv("SMOLMC_WL_V2") == nullptr)
        return h->lean_nslot == 2 ? smolmc_launch_wl_2(h, lp) : smolmc_launch_wl_4(h, lp);
    return h->lean_nslot == 2 ? smolmc_launch_lean_2(h, lp) : smolmc_launch_lean_4(h, lp);
}

static void free_samples(smolmc_handle *h) {
    for (SampleSlot &sl : h->slots) {
        if (sl.copy_done) { hipEventSynchronize(sl.copy_done); hipEventDestroy(sl.copy_done); }
        if (sl.kernel_done) hipEventDestroy(sl.kernel_done);
        if (sl.d) hipFree(sl.d);
        if (sl.hst) hipHostFree(sl.hst);
        sl = SampleSlot();
    }
    if (h->copy_stream) { hipStreamDestroy(h->copy_stream); h->copy_stream = nullptr; }
}

// parameter block of a universal-kernel launch: the handle's current KParams; on a lean handle (a
// replay the lean kernels cannot take) the potential field is used exactly when the lean kernel
// maintains it
static UParams univ_block_of(smolmc_handle *h) {
    UParams up = h->up;
    up.K = h->kp;
    if (h->lean && !h->lp.ew_field) up.K.ew_field = 0;
    return up;
}

static int run_steps(smolmc_handle *h, int64_t nsteps, const SampleBufs &smp) {
    // (per-bin feature statistics as sums for the lean Wang-Landau kernel and for mc_kernel with
    // update_period 1, as running means for the universal kernel)
    TRY(wl_set_representation(h, h->univ ? false : (h->lean ? h->lp.wl.sum_mode != 0 : h->kp.wl_sum_mode != 0)));
    if (h->univ) {
        UParams up = univ_block_of(h);
        up.K.steps_to_run = nsteps;
        up.K.smp = smp;
        return smolmc_launch_univ(h, up, 0);
    }
    if (h->lean) {
        // the lean kernels count steps in 32 bits: launches are split at 2^30 steps (on a
        // sample boundary when samples are being recorded)
        int64_t chunk = (int64_t)1 << 30;
        if (const char *c = getenv("SMOLMC_LAUNCH_CHUNK")) chunk = std::max<int64_t>(1, atoll(c)); // test hook
        if (smp.every) {
            if (smp.every > chunk) {
                if (getenv("SMOLMC_LAUNCH_CHUNK")) chunk = smp.every;
                else return fail("thin_by must be <= 2^30 steps");
            }
            chunk -= chunk % smp.every;
        }
        int64_t done = 0;
        while (done < nsteps) {
            const int64_t n = std::min(chunk, nsteps - done);
            LeanParams lp = h->lp;
            lp.smp = smp;
            if (smp.every) {
                const size_t rows = (size_t)(done / smp.every) * h->R;
                lp.smp.H += rows;
                lp.smp.feat += rows * (size_t)lp.F; // (lazy cluster features: rows of the scalar features)
                lp.smp.acc += rows;
                if (lp.smp.occ) lp.smp.occ += rows * h->Npad;
            }
            if (int rc = launch_lean(h, lp, n)) return rc;
            done += n;
        }
        if (is_lazy(h)) h->ce_dirty = true;
        return 0;
    }
    KParams kp = h->kp;
    kp.steps_to_run = nsteps;
    kp.smp = smp;
    return launch_mc(h, kp, 0);
}

extern "C" int smolmc_run(smolmc_handle *h, int64_t nsteps) {
    if (!h) return fail("null handle");
    if (nsteps < 0) return fail("nsteps must be non-negative");
    if (nsteps == 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    SampleBufs none;
    memset(&none, 0, sizeof(none));
    return run_steps(h, nsteps, none);
}

// ---- device-side sampling: a ring of two slots with asynchronous download ---------
// One sample of every walker from the handle's state arrays into row j of a slot -- for the kernels that do
// not record in-kernel (Wang-Landau: the per-walker L and L x F arrays; MCBias: the running bias): the
// block is then a sequence { launch of thin_by steps; this snapshot } queued on the stream, no host round trip.
struct SnapshotArgs {
    const uint8_t *occ; const double *features, *enthalpy; const uint8_t *last_acc; const double *bias;
    const double *wl_S; const long long *wl_hist, *wl_occ; const double *wl_mf, *wl_m;
    uint8_t *o_occ; double *o_feat, *o_H; uint8_t *o_acc; double *o_bias;
    double *o_wlS; long long *o_wlh, *o_wlo; double *o_wlf, *o_wlm;
    int R, F, L, Npad, mf_is_sums;
};
__global__ void __launch_bounds__(256) sample_snapshot_kernel(const SnapshotArgs A, long long j) {
    const int r = blockIdx.x, t = threadIdx.x;
    const size_t row = (size_t)j * A.R + r;
    if (A.o_occ) {
        const uint32_t *src = (const uint32_t *)(A.occ + (size_t)r * A.Npad);
        uint32_t *dst = (uint32_t *)(A.o_occ + row * A.Npad);
        for (int i = t; i < A.Npad / 4; i += 256) dst[i] = src[i];
    }
    for (int i = t; i < A.F; i += 256) A.o_feat[row * A.F + i] = A.features[(size_t)r * A.F + i];
    if (t == 0) {
        A.o_H[row] = A.enthalpy[r];
        A.o_acc[row] = A.last_acc[r];
        if (A.o_bias) A.o_bias[row] = A.bias[r];
        if (A.o_wlm) A.o_wlm[row] = A.wl_m[r];
    }
    if (A.o_wlS) {
        const size_t src = (size_t)r * A.L, dst = row * A.L;
        for (int i = t; i < A.L; i += 256) {
            A.o_wlS[dst + i] = A.wl_S[src + i];
            A.o_wlh[dst + i] = A.wl_hist[src + i];
            A.o_wlo[dst + i] = A.wl_occ[src + i];
        }
        // (mean = sum / occurrences where the kernel keeps per-bin SUMS, see wl_meanf_convert_kernel)
        for (size_t i = t; i < (size_t)A.L * A.F; i += 256) {
            double v = A.wl_mf[src * A.F + i];
            if (A.mf_is_sums) {
                const long long n = A.wl_occ[src + i / A.F];
                if (n > 0) v /= (double)n;
            }
            A.o_wlf[dst * A.F + i] = v;
        }
    }
}

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int smolmc_run_sampled(smolmc_handle *h, int64_t nsamples, int64_t thin_by, int flags) {
    if (!h) return fail("null handle");
    if (nsamples <= 0 || thin_by <= 0) return fail("nsamples and thin_by must be positive");
    HIPCHK(hipSetDevice(h->device));
    const bool wl = h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU;
    if ((flags & SMOLMC_SAMPLE_BIAS) && !h->kp.bias_type) return fail("the model has no bias term");
    if ((flags & SMOLMC_SAMPLE_WL) && !wl) return fail("handle is not a Wang-Landau kernel");
    if (thin_by > ((int64_t)1 << 30) && h->lean && getenv("SMOLMC_LAUNCH_CHUNK") == nullptr)
        return fail("thin_by must be <= 2^30 steps"); // (before a slot is touched: run_steps would refuse it)
    if (!h->copy_stream) HIPCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    // The slot this block goes to is the older one.  When it still holds a block nobody fetched, so does the other
    // (it is younger): the ring is full and the call is refused -- ABI 7 dropped the older block here without a word.
    SampleSlot &sl = h->slots[h->next_slot];
    if (sl.state == 1) {
        fail("sample ring full: both slots hold blocks that were not fetched (call smolmc_get_samples* or "
             "smolmc_discard_samples first)");
        return SMOLMC_ERR_RING_FULL;
    }
    if (!sl.kernel_done) {
        HIPCHK(hipEventCreateWithFlags(&sl.kernel_done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.copy_done, hipEventDisableTiming));
    }
    // layout of the block inside the slot's arenas: worked out on a COPY of the slot record, which replaces the
    // record only when every launch of the block has been queued -- a call that fails on the way (an allocation, a
    // launch) leaves the ring as it was: the slot keeps its delivered block, next_slot does not move
    SampleSlot w = sl;
    const size_t rows = (size_t)nsamples * h->R, F = (size_t)h->F, L = (size_t)h->L;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += up256(bytes); return o; };
    w.o_H = take(rows * 8);
    w.o_feat = take(rows * F * 8);
    w.o_acc = take(rows);
    w.o_bias = (flags & SMOLMC_SAMPLE_BIAS) ? take(rows * 8) : 0;
    w.o_wlm = w.o_wlS = w.o_wlh = w.o_wlo = w.o_wlf = 0;
    if (flags & SMOLMC_SAMPLE_WL) {
        w.o_wlm = take(rows * 8);
        w.o_wlS = take(rows * L * 8);
        w.o_wlh = take(rows * L * 8);
        w.o_wlo = take(rows * L * 8);
        w.o_wlf = take(rows * L * F * 8);
    }
    w.o_occ = (flags & SMOLMC_SAMPLE_OCCUPANCY) ? take(rows * h->Npad) : 0;
    // lazy cluster features, rows recorded in-kernel: the kernels write rows of scalar features and the occupancy of
    // every sample; the cluster features of the rows are evaluated from those when the launch is through.  What the
    // caller did not ask for sits behind the part of the arena that is downloaded.
    // Biased Metropolis handles on the lean families record their rows in-kernel like the unbiased ones (round 6: the
    // running bias is a scalar the kernel carries anyway; LeanParams::smp_bias_off tells it where the column is).  The
    // snapshot path -- one launch + one snapshot kernel per sample -- is left to Wang-Landau (per-walker L and L x F
    // arrays) and to biased handles on mc_kernel / the universal kernel.  SMOLMC_NO_INKERNEL_BIAS: A/B switch.
    const bool inkernel_bias = (flags & SMOLMC_SAMPLE_BIAS) && !(flags & SMOLMC_SAMPLE_WL) && h->lean && !h->univ &&
                               h->lp.bias_type && getenv("SMOLMC_NO_INKERNEL_BIAS") == nullptr;
    const bool snapshot_path = (flags & SMOLMC_SAMPLE_WL) || ((flags & SMOLMC_SAMPLE_BIAS) && !inkernel_bias);
    const bool lazy_rows = is_lazy(h) && !snapshot_path;
    size_t download = at, o_scal = 0, o_occ_int = w.o_occ;
    if (lazy_rows) {
        download = at;
        o_scal = take(rows * (size_t)std::max(1, lazy_nscal(h)) * 8);
        if (!(flags & SMOLMC_SAMPLE_OCCUPANCY)) o_occ_int = take(rows * h->Npad);
    }
    if (lazy_rows && rows > 0x7fffffffull) return fail("lazy cluster features: more than 2^31 sample rows in one block");
    // the slot's previous block (delivered, or none) must have left the device before its arenas are reused
    if (sl.state != 0) HIPCHK(hipEventSynchronize(sl.copy_done));
    if (at > sl.cap) {
        // (growing frees the delivered block's mirror: the slot is empty from here on, whatever happens next)
        sl.state = 0;
        if (sl.d) hipFree(sl.d);
        if (sl.hst) hipHostFree(sl.hst);
        sl.d = sl.hst = nullptr;
        sl.cap = 0;
        HIPCHK(hipMalloc((void **)&sl.d, at));
        if (hipHostMalloc((void **)&sl.hst, at, hipHostMallocDefault) != hipSuccess) {
            hipFree(sl.d);
            sl.d = nullptr;
            return fail("hipHostMalloc failed (pinned mirror of the sample ring)");
        }
        sl.cap = at;
        w.d = sl.d; w.hst = sl.hst; w.cap = sl.cap;
    }
    w.used = lazy_rows ? download : at;
    w.n = nsamples;
    w.flags = flags;
    SampleBufs smp;
    memset(&smp, 0, sizeof(smp));
    smp.every = thin_by;
    smp.H = (double *)(sl.d + w.o_H);
    smp.feat = (double *)(sl.d + (lazy_rows ? o_scal : w.o_feat));
    smp.acc = sl.d + w.o_acc;
    smp.occ = (flags & SMOLMC_SAMPLE_OCCUPANCY) || lazy_rows ? sl.d + o_occ_int : nullptr;
    if (!snapshot_path) {
        if (inkernel_bias) {
            if ((w.o_bias - w.o_H) / 8 > 0xffffffffull) return fail("sample block too large for the in-kernel bias column");
            h->lp.smp_bias_off = (uint32_t)((w.o_bias - w.o_H) / 8);
        }
        const int rc_run = run_steps(h, nsamples * thin_by, smp); // the kernels record the rows themselves, one launch
        h->lp.smp_bias_off = 0;
        if (rc_run) return rc_run;
        if (lazy_rows) {
            TRY(launch_eval_full(h, sl.d + o_occ_int, (int)rows, (double *)(sl.d + w.o_feat), 1));
            TRY(lazy_scalars(h, (double *)(sl.d + w.o_feat), (double *)(sl.d + o_scal), rows, 0));
        }
    } else {
        SampleBufs none;
        memset(&none, 0, sizeof(none));
        KParams &kp = h->kp;
        SnapshotArgs A;
        memset(&A, 0, sizeof(A));
        A.occ = kp.occ; A.features = kp.features; A.enthalpy = kp.enthalpy; A.last_acc = kp.last_acc; A.bias = kp.bias;
        A.wl_S = kp.wl_entropy; A.wl_hist = kp.wl_hist; A.wl_occ = kp.wl_occur; A.wl_mf = kp.wl_meanf; A.wl_m = kp.wl_m;
        A.o_occ = smp.occ; A.o_feat = smp.feat; A.o_H = smp.H; A.o_acc = smp.acc;
        A.o_bias = (flags & SMOLMC_SAMPLE_BIAS) ? (double *)(sl.d + w.o_bias) : nullptr;
        if (flags & SMOLMC_SAMPLE_WL) {
            A.o_wlm = (double *)(sl.d + w.o_wlm); A.o_wlS = (double *)(sl.d + w.o_wlS);
            A.o_wlh = (long long *)(sl.d + w.o_wlh); A.o_wlo = (long long *)(sl.d + w.o_wlo);
            A.o_wlf = (double *)(sl.d + w.o_wlf);
        }
        A.R = h->R; A.F = h->F; A.L = h->L; A.Npad = h->Npad;
        for (int64_t j = 0; j < nsamples; ++j) {
            TRY(run_steps(h, thin_by, none));
            TRY(ensure_features(h)); // (lazy cluster features: the snapshot reads kp.features)
            A.mf_is_sums = h->wl_sums ? 1 : 0; // (the representation the launch left the per-bin statistics in)
            hipLaunchKernelGGL(sample_snapshot_kernel, dim3((unsigned)h->R), dim3(256), 0, h->stream, A, (long long)j);
            HIPCHK(hipGetLastError());
        }
    }
    // download on the copy stream as soon as the block is complete; the next block's launches do not wait for it
    // (the mirror is about to be overwritten: from here the slot no longer holds its old block)
    sl.state = 0;
    HIPCHK(hipEventRecord(sl.kernel_done, h->stream));
    HIPCHK(hipStreamWaitEvent(h->copy_stream, sl.kernel_done, 0));
    HIPCHK(hipMemcpyAsync(sl.hst, sl.d, w.used, hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(hipEventRecord(sl.copy_done, h->copy_stream));
    // (the next writer of this slot's arenas -- the call after next -- waits for copy_done above)
    // every launch is queued: the block takes the slot
    w.state = 1;
    w.seq = ++h->slot_seq;
    sl = w;
    h->next_slot ^= 1;
    return 0;
}

// The block the next smolmc_get_samples* call delivers (ABI 8): its sample count and flags, so that a caller can size
// its arrays from the ring itself -- get_samples takes no lengths -- and the number of blocks queued and not fetched.
static SampleSlot *next_delivery(smolmc_handle *h) {
    SampleSlot *sl = nullptr;
    for (SampleSlot &c : h->slots)
        if (c.state == 1 && (!sl || c.seq < sl->seq)) sl = &c;
    if (!sl)
        for (SampleSlot &c : h->slots)
            if (c.state == 2 && (!sl || c.seq > sl->seq)) sl = &c;
    return sl;
}
extern "C" int smolmc_pending_samples(smolmc_handle *h, int *n_pending, int64_t *nsamples, int *flags) {
    if (!h) return fail("null handle");
    int np = 0;
    for (SampleSlot &c : h->slots) np += c.state == 1;
    const SampleSlot *sl = next_delivery(h);
    if (n_pending) *n_pending = np;
    if (nsamples) *nsamples = sl ? (int64_t)sl->n : 0;
    if (flags) *flags = sl ? sl->flags : 0;
    return 0;
}
// Forget every block of the ring, fetched or not (a sampling loop that was abandoned half way: the blocks it queued
// must not be delivered to the next one).  The walkers keep the state the queued launches leave them in.
extern "C" int smolmc_discard_samples(smolmc_handle *h) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    for (SampleSlot &c : h->slots) {
        if (c.state != 0 && c.copy_done) HIPCHK(hipEventSynchronize(c.copy_done));
        c.state = 0;
    }
    return 0;
}

// host side of a delivery: pinned mirror -> the caller's arrays, a few threads for the big ones (the copy of a
// block overlaps the kernel of the next one as long as it is shorter)
static void host_copy(void *dst, const void *src, size_t bytes) {
    const size_t big = (size_t)8 << 20;
    if (bytes < big) { memcpy(dst, src, bytes); return; }
    const int nt = 4;
    std::thread th[nt];
    const size_t part = (bytes / nt + 63) & ~(size_t)63;
    for (int k = 0; k < nt; ++k) {
        const size_t a = std::min(bytes, part * k), b = std::min(bytes, part * (k + 1));
        th[k] = std::thread([=]() { memcpy((char *)dst + a, (const char *)src + a, b - a); });
    }
    for (int k = 0; k < nt; ++k) th[k].join();
}
// occupancy rows: Npad bytes apart in the ring, N bytes (uint8) or N int32 in the caller's array
// (new_of != nullptr: a relabelled handle -- column p of the caller's row is byte new_of[p] of the ring's)
template <typename T> static void host_copy_occ(T *dst, const uint8_t *src, size_t rows, int N, int Npad, const int32_t *new_of) {
    auto work = [=](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            const uint8_t *s = src + r * Npad;
            T *d = dst + r * (size_t)N;
            if (new_of) for (int i = 0; i < N; ++i) d[i] = (T)s[new_of[i]];
            else if (sizeof(T) == 1) memcpy(d, s, (size_t)N);
            else for (int i = 0; i < N; ++i) d[i] = (T)s[i];
        }
    };
    if (rows * (size_t)N < ((size_t)8 << 20)) { work(0, rows); return; }
    const int nt = 4;
    std::thread th[nt];
    for (int k = 0; k < nt; ++k) th[k] = std::thread(work, rows * k / nt, rows * (k + 1) / nt);
    for (int k = 0; k < nt; ++k) th[k].join();
}

static int get_samples_impl(smolmc_handle *h, double *enthalpy, double *features, uint8_t *accepted,
                            int32_t *occ32, uint8_t *occ8, double *bias, double *wl_S, int64_t *wl_hist,
                            int64_t *wl_occ, double *wl_mf, double *wl_m) {
    if (!h) return fail("null handle");
    // the oldest block not yet delivered; with none pending, the newest delivered one again
    SampleSlot *sl = next_delivery(h);
    if (!sl) return fail("no samples recorded: call smolmc_run_sampled first");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(sl->copy_done));
    const size_t rows = (size_t)sl->n * h->R, F = (size_t)h->F, L = (size_t)h->L;
    if ((occ32 || occ8) && !(sl->flags & SMOLMC_SAMPLE_OCCUPANCY)) return fail("occupancies were not recorded (flags bit 0)");
    if (bias && !(sl->flags & SMOLMC_SAMPLE_BIAS)) return fail("the bias was not recorded (flags bit 1)");
    if ((wl_S || wl_hist || wl_occ || wl_mf || wl_m) && !(sl->flags & SMOLMC_SAMPLE_WL))
        return fail("the Wang-Landau trace was not recorded (flags bit 2)");
    if (enthalpy) host_copy(enthalpy, sl->hst + sl->o_H, rows * 8);
    if (features) host_copy(features, sl->hst + sl->o_feat, rows * F * 8);
    if (accepted) host_copy(accepted, sl->hst + sl->o_acc, rows);
    if (bias) host_copy(bias, sl->hst + sl->o_bias, rows * 8);
    if (wl_m) host_copy(wl_m, sl->hst + sl->o_wlm, rows * 8);
    if (wl_S) host_copy(wl_S, sl->hst + sl->o_wlS, rows * L * 8);
    if (wl_hist) host_copy(wl_hist, sl->hst + sl->o_wlh, rows * L * 8);
    if (wl_occ) host_copy(wl_occ, sl->hst + sl->o_wlo, rows * L * 8);
    if (wl_mf) host_copy(wl_mf, sl->hst + sl->o_wlf, rows * L * F * 8);
    const int32_t *new_of = h->relabelled ? h->new_of.data() : nullptr;
    if (occ8) host_copy_occ<uint8_t>(occ8, sl->hst + sl->o_occ, rows, h->N, h->Npad, new_of);
    if (occ32) host_copy_occ<int32_t>(occ32, sl->hst + sl->o_occ, rows, h->N, h->Npad, new_of);
    sl->state = 2;
    return 0;
}
extern "C" int smolmc_get_samples(smolmc_handle *h, double *enthalpy, double *features,
                                  uint8_t *accepted, int32_t *occupancy) {
    return get_samples_impl(h, enthalpy, features, accepted, occupancy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}
extern "C" int smolmc_get_samples_u8(smolmc_handle *h, double *enthalpy, double *features,
                                     uint8_t *accepted, uint8_t *occupancy) {
    return get_samples_impl(h, enthalpy, features, accepted, nullptr, occupancy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}
extern "C" int smolmc_get_samples_ex(smolmc_handle *h, double *enthalpy, double *features, uint8_t *accepted,
                                     uint8_t *occupancy_u8, double *bias, double *wl_entropy, int64_t *wl_histogram,
                                     int64_t *wl_occurrences, double *wl_mean_features, double *wl_mod_factor) {
    return get_samples_impl(h, enthalpy, features, accepted, nullptr, occupancy_u8, bias, wl_entropy, wl_histogram,
                            wl_occurrences, wl_mean_features, wl_mod_factor);
}

// lean replay kernels that exist: Metropolis flips / swaps (plain, KF, with MCBias), multi-sublattice,
// Wang-Landau; TableFlip handles have their own (smolmc_table_replay_available)
static bool smolmc_lean_replay_takes(const smolmc_handle *h) {
    if (h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP) return false;
    if (h->lean_multi_wl && h->lean_kf) return false; // (no replay instantiation of the KFW kernel: mc_kernel / the universal kernel)
    if (h->lp.bias_type) return SMOLMC_HAVE_BIAS_REPLAY != 0;
    return true;
}
static bool smolmc_table_replay_available() { return SMOLMC_HAVE_TABLE_REPLAY != 0; }
static int smolmc_launch_lean_replay(smolmc_handle *h, const LeanParams &lp) {
    const bool wl = h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU;
#if SMOLMC_HAVE_TABLE_REPLAY
    if (h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP) {
        LeanParams q = lp;
        TRY(update_walker_order(h, q));
        if (h->lean_multi)
            return h->lean_nslot == 2 ? smolmc_launch_multi_table_replay_2(h, q)
                                      : (h->lean_nslot == 4 ? smolmc_launch_multi_table_replay_4(h, q) : smolmc_launch_multi_table_replay_8(h, q));
        return h->lean_nslot == 2 ? smolmc_launch_table_replay_2(h, q) : smolmc_launch_table_replay_4(h, q);
    }
#endif
#if SMOLMC_HAVE_BIAS_REPLAY
    if (lp.bias_type && h->lean_multi)
        return h->lean_nslot == 2 ? smolmc_launch_multi_bias_replay_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_bias_replay_4(h, lp) : smolmc_launch_multi_bias_replay_8(h, lp));
    if (lp.bias_type) return h->lean_nslot == 2 ? smolmc_launch_lean_bias_replay_2(h, lp) : smolmc_launch_lean_bias_replay_4(h, lp);
#endif
    if (h->lean_multi_wl)
        return h->lean_nslot == 2 ? smolmc_launch_multi_wl_replay_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_wl_replay_4(h, lp) : smolmc_launch_multi_wl_replay_8(h, lp));
    if (h->lean_multi)
        return h->lean_nslot == 2 ? smolmc_launch_multi_replay_2(h, lp)
                                  : (h->lean_nslot == 4 ? smolmc_launch_multi_replay_4(h, lp) : smolmc_launch_multi_replay_8(h, lp));
    return wl ? (h->lean_nslot == 2 ? smolmc_launch_wl_replay_2(h, lp) : smolmc_launch_wl_replay_4(h, lp))
              : (h->lean_nslot == 2 ? smolmc_launch_lean_replay_2(h, lp) : smolmc_launch_lean_replay_4(h, lp));
}

extern "C" int smolmc_replay(smolmc_handle *h, int64_t nsteps, const int32_t *steps, const double *uniforms,
                             const double *log_priori, uint8_t *accepted_out, double *enthalpy_out,
                             double *log_priori_out) {
    if (!h || !steps || !uniforms) return fail("null argument");
    if (nsteps <= 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    const size_t n = (size_t)h->R * nsteps;
    std::vector<int32_t> steps_engine;
    steps = steps_in(h, steps, n, steps_engine);
    // every flip: a changeable site of an active sublattice (the lean kernels index their tables
    // relative to the active range; a code beyond the site's species would index past its tensors)
    int max_flips = 0;
    bool repeated_site = false; // a record that flips one site twice (sequential-flip semantics, expansion.py:217-229)
    for (size_t i = 0; i < n; ++i) {
        const int32_t *st = steps + i * SMOLMC_STEP_ROW;
        int nf = 0;
        for (; nf < SMOLMC_MAX_STEP_FLIPS && st[2 * nf] >= 0; ++nf) {
            const int32_t site = st[2 * nf], code = st[2 * nf + 1];
            if (site >= h->N) return fail("replay step out of range");
            if (!h->site_active[site]) return fail("replay step out of range: the site is not changeable (not on an active sublattice)");
            if (code < 0 || code >= (int)h->site_ncodes[site]) return fail("replay step out of range (species code)");
            for (int g = 0; g < nf; ++g) repeated_site |= st[2 * g] == site;
        }
        max_flips = std::max(max_flips, nf);
    }
    bool priori_given = false; // a factor the Flip / Swap kernels do not model
    if (log_priori)
        for (size_t i = 0; i < n && !priori_given; ++i) priori_given = log_priori[i] == log_priori[i] && log_priori[i] != 0.0;
    const bool table = h->cfg.step_type == SMOLMC_STEP_TABLE_FLIP;
    // Lean handles replay on their own kernels (REPLAY instantiations of mc_lean_kernel,
    // mc_lean_multi_kernel, mc_wl_kernel, mc_table_kernel, mc_table_multi_kernel); steps those cannot
    // take (more than two flips on a Flip / Swap handle, a given a-priori factor there), universal
    // handles and SMOLMC_REPLAY_UNIVERSAL take the universal kernel; SMOLMC_REPLAY_GENERAL: mc_kernel.
    const bool two_flip_ok = max_flips <= 2 && !priori_given && !table;
    const bool want_general = getenv("SMOLMC_REPLAY_GENERAL") != nullptr && two_flip_ok && h->general_ok;
    // (the lean TableFlip replay kernels pick sites without replacement like the usher: a record with a repeated
    // site -- valid for the boundary -- takes the universal kernel, which evaluates it flip by flip)
    // (a biased or Wang-Landau TableFlip handle has no REPLAY instantiation: the universal kernel replays its records)
    const bool lean_table_replay = h->lean && table && smolmc_table_replay_available() && nsteps < ((int64_t)1 << 30) && !repeated_site &&
                                   !h->lp.bias_type && h->cfg.kernel_type != SMOLMC_KERNEL_WANGLANDAU;
    // (a Flip handle's own kernel takes single flips: records of two flips go to mc_kernel / the universal kernel)
    const bool lean_shape_ok = two_flip_ok && (h->cfg.step_type != SMOLMC_STEP_FLIP || max_flips <= 1);
    const bool lean_replay = h->lean && !want_general && nsteps < ((int64_t)1 << 30) && getenv("SMOLMC_REPLAY_UNIVERSAL") == nullptr &&
                             ((lean_shape_ok && smolmc_lean_replay_takes(h)) || lean_table_replay);
    const bool general_replay = !lean_replay && !h->univ && two_flip_ok && h->general_ok && getenv("SMOLMC_REPLAY_UNIVERSAL") == nullptr;
    if (getenv("SMOLMC_DEBUG"))
        fprintf(stderr, "[smolmc] replay path=%s max_flips=%d priori_given=%d\n",
                lean_replay ? (table ? "lean-table" : "lean") : (general_replay ? "general" : "universal"), max_flips, (int)priori_given);
    int *d_steps = nullptr, *d_err = nullptr;
    double *d_u = nullptr, *d_H = nullptr, *d_lp = nullptr, *d_lpo = nullptr;
    uint8_t *d_acc = nullptr;
    // the two-flip kernels read records of four ints
    std::vector<int32_t> packed;
    const bool narrow = (lean_replay && !table) || general_replay;
    if (narrow) {
        packed.resize(n * 4);
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 4; ++k) packed[i * 4 + k] = steps[i * SMOLMC_STEP_ROW + k];
    }
    const size_t row_bytes = narrow ? 16 : SMOLMC_STEP_ROW * 4;
    hipError_t e = hipMalloc((void **)&d_steps, n * row_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_err, 16);
    if (e == hipSuccess) e = hipMemset(d_err, 0, 16);
    if (e == hipSuccess) e = hipMalloc((void **)&d_u, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_H, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_acc, n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_lpo, n * 8);
    if (e == hipSuccess) e = hipMemset(d_lpo, 0, n * 8);
    if (e == hipSuccess && log_priori) {
        e = hipMalloc((void **)&d_lp, n * 8);
        if (e == hipSuccess) e = hipMemcpy(d_lp, log_priori, n * 8, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(d_steps, narrow ? packed.data() : steps, n * row_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_u, uniforms, n * 8, hipMemcpyHostToDevice);
    int rc = 0;
    if (e == hipSuccess && lean_replay) {
        const bool wl = h->cfg.kernel_type == SMOLMC_KERNEL_WANGLANDAU;
        rc = wl_set_representation(h, wl && h->lp.wl.sum_mode);
        LeanParams lp = h->lp;
        memset(&lp.smp, 0, sizeof(lp.smp));
        lp.steps = nsteps;
        lp.rp_steps = d_steps; lp.rp_u = d_u; lp.rp_acc = d_acc; lp.rp_H = d_H; lp.rp_err = d_err;
        lp.rp_lp = d_lp; lp.rp_lp_out = d_lpo;
        if (!rc) rc = smolmc_launch_lean_replay(h, lp);
        if (is_lazy(h)) h->ce_dirty = true;
        if (!rc) e = hipStreamSynchronize(h->stream);
        int bad = 0;
        if (!rc && e == hipSuccess) e = hipMemcpy(&bad, d_err, 4, hipMemcpyDeviceToHost);
        if (!rc && e == hipSuccess && bad)
            rc = table ? fail("replay: Step is not in flip table (neither a canonical swap nor +-(a row of flip_table)); "
                              "the walkers have been advanced -- set the state again")
                       : fail("replay step does not fit the handle's step type (a swap handle takes proper swaps: "
                              "code1 == species at site2, code2 == species at site1; a flip handle single flips); "
                              "the walkers have been advanced -- set the state again");
    } else if (e == hipSuccess && general_replay) {
        rc = wl_set_representation(h, h->kp.wl_sum_mode != 0);
        if (!rc) rc = ensure_features(h); // (mc_kernel updates the features incrementally)
        KParams kp = h->kp;
        kp.steps_to_run = nsteps;
        kp.rp_steps = d_steps;
        kp.rp_u = d_u;
        kp.rp_acc = d_acc;
        kp.rp_H = d_H;
        if (!rc) rc = launch_mc(h, kp, 1);
        if (!rc) rc = lazy_scalars_from_features(h);
        if (!rc) e = hipStreamSynchronize(h->stream);
    } else if (e == hipSuccess) {
        rc = wl_set_representation(h, false);
        if (!rc) rc = ensure_features(h); // (the universal kernel updates the features incrementally)
        UParams up = univ_block_of(h);
        up.K.steps_to_run = nsteps;
        memset(&up.K.smp, 0, sizeof(up.K.smp));
        up.K.rp_steps = d_steps; up.K.rp_u = d_u; up.K.rp_acc = d_acc; up.K.rp_H = d_H;
        up.rp_lp = d_lp; up.rp_lp_out = d_lpo; up.rp_err = d_err;
        if (!rc) rc = smolmc_launch_univ(h, up, 1);
        if (!rc) rc = lazy_scalars_from_features(h);
        if (!rc) e = hipStreamSynchronize(h->stream);
        int bad = 0;
        if (!rc && e == hipSuccess) e = hipMemcpy(&bad, d_err, 4, hipMemcpyDeviceToHost);
        if (!rc && e == hipSuccess && bad)
            rc = fail("replay: Step is not in flip table (neither a canonical swap nor +-(a row of flip_table)); "
                      "the walkers have been advanced -- set the state again");
    }
    if (!rc && e == hipSuccess && accepted_out) e = hipMemcpy(accepted_out, d_acc, n, hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && enthalpy_out) e = hipMemcpy(enthalpy_out, d_H, n * 8, hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && log_priori_out) e = hipMemcpy(log_priori_out, d_lpo, n * 8, hipMemcpyDeviceToHost);
    hipFree(d_steps);
    hipFree(d_err);
    hipFree(d_u);
    hipFree(d_H);
    hipFree(d_acc);
    hipFree(d_lpo);
    if (d_lp) hipFree(d_lp);
    if (rc) return rc;
    if (e != hipSuccess) return fail(std::string("replay: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int smolmc_last_kernel_ms(smolmc_handle *h, float *ms) {
    if (!h || !ms) return fail("null argument");
    if (!h->timed) return fail("no kernel has been launched yet");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}

static int ensure_eval_occ(smolmc_handle *h, size_t nocc) {
    const size_t need = nocc * h->Npad;
    if (need > h->eval_occ_cap) {
        if (h->d_eval_occ) hipFree(h->d_eval_occ);
        h->d_eval_occ = nullptr;
        h->eval_occ_cap = 0;
        HIPCHK(hipMalloc((void **)&h->d_eval_occ, need));
        h->eval_occ_cap = need;
    }
    return 0;
}

extern "C" int smolmc_eval_full(smolmc_handle *h, const int32_t *occ, int nocc, double *features) {
    if (!h || !occ || !features) return fail("null argument");
    if (nocc <= 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    TRY(ensure_eval_occ(h, nocc));
    std::vector<int32_t> occ_engine;
    occ = occ_in(h, occ, (size_t)nocc, occ_engine);
    TRY(upload_occ(h, occ, nocc, h->d_eval_occ));
    double *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_out, (size_t)nocc * h->F * 8));
    int rc = launch_eval_full(h, h->d_eval_occ, nocc, d_out);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (!rc && e == hipSuccess) e = hipMemcpy(features, d_out, (size_t)nocc * h->F * 8, hipMemcpyDeviceToHost);
    hipFree(d_out);
    if (rc) return rc;
    if (e != hipSuccess) return fail(std::string("eval_full: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int smolmc_eval_delta(smolmc_handle *h, const int32_t *occ, const int32_t *flips, int nstep,
                                 double *dfeatures) {
    if (!h || !occ || !flips || !dfeatures) return fail("null argument");
    if (nstep <= 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    std::vector<int32_t> occ_engine, flips_engine;
    occ = occ_in(h, occ, 1, occ_engine);
    flips = steps_in(h, flips, (size_t)nstep, flips_engine);
    for (int i = 0; i < nstep; ++i)
        for (int f = 0; f < SMOLMC_MAX_STEP_FLIPS; ++f) {
            const int s = flips[(size_t)i * SMOLMC_STEP_ROW + 2 * f], c = flips[(size_t)i * SMOLMC_STEP_ROW + 2 * f + 1];
            if (s < 0) break;
            if (s >= h->N || c < 0 || c >= (int)h->site_ncodes[s]) return fail("flip out of range");
        }
    TRY(ensure_eval_occ(h, 1));
    TRY(upload_occ(h, occ, 1, h->d_eval_occ));
    int *d_fl = nullptr;
    double *d_out = nullptr;
    hipError_t e = hipMalloc((void **)&d_fl, (size_t)nstep * SMOLMC_STEP_ROW * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)nstep * h->F * 8);
    if (e == hipSuccess) e = hipMemcpy(d_fl, flips, (size_t)nstep * SMOLMC_STEP_ROW * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(eval_delta_kernel, dim3(nstep), dim3(64), 0, h->stream, h->rt, h->d_eval_occ,
                           d_fl, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(dfeatures, d_out, (size_t)nstep * h->F * 8, hipMemcpyDeviceToHost);
    hipFree(d_fl);
    hipFree(d_out);
    if (e != hipSuccess) return fail(std::string("eval_delta: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int smolmc_export_enthalpy_dev(smolmc_handle *h, double *dst_dev) {
    if (!h || !dst_dev) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(dst_dev, h->kp.enthalpy, (size_t)h->R * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

__global__ void beta_from_T_kernel(const double *T, double *beta, int R, double kB) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) beta[r] = 1.0 / (kB * T[r]);
}

// One exchange attempt of a temperature ladder decided on the device: the serial section of the N-rank config-5 loop
// (host NumPy over the all-gathered enthalpies until round 5: parallel.ReplicaExchange.decide).  One workgroup; the
// rung -> walker map lives in LDS.  Every operation is the NumPy path's own (beta = 1 / (kB T), the product of the two
// differences, the comparison with the host-made log u): the same decisions bit for bit.
__global__ void __launch_bounds__(1024) rex_decide_kernel(const int n, const int first, const int parity, const double *H,
                                                          const double *ladder, const double *log_u, int *rung_of,
                                                          long long *stats, double *beta_out, const int R, const double kB) {
    extern __shared__ int walker_at[]; // [n] rung -> walker
    for (int g = threadIdx.x; g < n; g += blockDim.x) walker_at[rung_of[g]] = g;
    __syncthreads();
    const int npairs = (n - 1 - parity + 1) / 2; // pairs (k, k + 1), k = parity, parity + 2, ... < n - 1
    for (int p = threadIdx.x; p < npairs; p += blockDim.x) {
        const int k = parity + 2 * p;
        const int a = walker_at[k], b = walker_at[k + 1];
        const double bk = 1.0 / (kB * ladder[k]), bk1 = 1.0 / (kB * ladder[k + 1]);
        const double expo = (bk - bk1) * (H[a] - H[b]);
        const bool acc = expo >= 0.0 || log_u[p] < expo;
        if (stats) {
            stats[k] += 1;                          // attempted
            if (acc) stats[(n - 1) + k] += 1;       // accepted
        }
        if (acc) { rung_of[a] = k + 1; rung_of[b] = k; } // (the pairs of one parity are disjoint)
    }
    __threadfence_block();
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) beta_out[r] = 1.0 / (kB * ladder[rung_of[first + r]]);
}

extern "C" int smolmc_exchange_dev(smolmc_handle *h, int n_total, int first, int parity, const double *enthalpy_all_dev,
                                   const double *ladder_dev, const double *log_u_dev, int32_t *rung_of_dev,
                                   int64_t *stats_dev) {
    if (!h || !enthalpy_all_dev || !ladder_dev || !log_u_dev || !rung_of_dev) return fail("null argument");
    if (n_total < 2 || n_total > 16384) return fail("exchange ladder must hold 2 .. 16384 walkers (the rung map lives in LDS)");
    if (first < 0 || first + h->R > n_total) return fail("this handle's walkers are out of range of the ladder");
    if (parity != 0 && parity != 1) return fail("parity must be 0 or 1");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(rex_decide_kernel, dim3(1), dim3(1024), (size_t)n_total * sizeof(int), h->stream, n_total, first, parity,
                       enthalpy_all_dev, ladder_dev, log_u_dev, (int *)rung_of_dev, (long long *)stats_dev, h->d_beta, h->R,
                       SMOLMC_KB);
    HIPCHK(hipGetLastError());
    h->order_dirty = true; // (the launch order of the TableFlip kernels follows the temperatures)
    return 0;
}

extern "C" int smolmc_import_temperature_dev(smolmc_handle *h, const double *src_dev) {
    if (!h || !src_dev) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(beta_from_T_kernel, dim3((h->R + 63) / 64), dim3(64), 0, h->stream, src_dev,
                       h->d_beta, h->R, SMOLMC_KB);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->order_dirty = true;
    return 0;
}
