// ggml_flash_attn_ext for Whisper's head_dim 64 (ggml/src/ggml.c:5418-5460; CPU reference ggml-cpu/ops.cpp:8479-8715):
//     dst[:, h, t] = softmax_k( scale * <f16(q[:, t, h]), k[:, k, h]> + mask[k, t] ) . v[:, k, h]
// q F32 (rounded to f16 like the CPU's q_to_vec_dot), k/v F16, mask F16 or none, f32 accumulation throughout.
//
//   k_fattn_mfma : T > 8 (encoder 1500 x 1536, prompt).  128 queries x 1 head per workgroup, 4 waves x 32 queries per key group
//                  (r03: NG key groups per workgroup, merged through LDS — see the kernel),
//                  64-key tiles staged once per workgroup in LDS (K row-major XOR-swizzled, V transposed with a
//                  conflict-free 136-byte row pitch).  S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_32x32x16_f16, so a
//                  lane owns ONE query column: softmax statistics are per-lane scalars (one shuffle with lane^32),
//                  the P fragment feeds the second MFMA straight from registers (the k-index permutation inside the
//                  MFMA is applied identically to V^T), and the O rescale is a per-lane scalar multiply.
//   T <= 8 (decoder step) goes to k_fattn_dec in decode.hip (128-key partial records, merged by the consumer).
#include "common.h"
#include "qrows.h"
#include <math.h>
#include <stdlib.h>

#define FA_D 64

struct FattnArgs {
    dtensor q, k, v, m, d;
    int has_mask; float scale;
    int T, n_kv, H, rk2, rv2;
    uint16_t * prep; int prep_ld;      // also write k_prep_act(mode 1) of the result seen as [T][H*64]: the O-projection's activations
    int8_t * rq; float * rd;           // ... or its Q8_0 ROWS (k_prep_act mode 3, qrows.h): the int8 tile GEMM's activations
};

// ---------------------------------------------------------------------------------------------------
// MFMA kernel
// ---------------------------------------------------------------------------------------------------
#define KT 64                 // keys per tile
#define VT_PITCH 136          // bytes per d-row of the transposed V tile (64 keys * 2 B + 8 B pad)

__device__ __forceinline__ int k_off(int row, int slot) { return row*128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// NG key groups of 4 waves each share the 128 queries of the workgroup: group g walks the contiguous tile range
// [g*tpg, (g+1)*tpg) of the keys with its own LDS tile pair, and the groups' (m, l, O) records are merged through LDS at the end
// (group 0 writes the result).  1500 queries x 20 heads are only 938 waves of 32 queries for 1024 SIMDs: with one group every
// SIMD runs ONE wave whose LDS reads, MFMAs and softmax VALU work are strictly serial; with NG groups a SIMD holds NG waves
// whose MFMA and VALU segments overlap.  MASK = false (the encoder) folds scale and log2(e) into one fma per score:
// p = exp2(s*c - m*c), c = scale*log2(e), running maximum kept on the raw scores (scale > 0 is checked on the host).
// VTR: the V tile stays ROW-major in LDS ([key][64 d], 192-byte pitch, two ds_write_b128 per thread and tile) and the P.V fragment
// is gathered by ds_read_b64_tr_b16, gfx950's transposing LDS read: lane i of a 16-lane group supplies the 8-byte chunk
// (row i/4, columns 4(i%4)...) of a 4 x 16 block at any row pitch and receives column i of it (scripts/probes/tr_b16_probe.hip,
// profiles/r03b_tr_b16_probe.txt) — the two 4-key column segments per lane the fragment consists of.  The 192-byte pitch puts the
// four rows of a 32-lane half on disjoint bank spans (0 / 48 / 32 / 16 dwords mod 64).  !VTR: V transposed on its way into LDS by
// 16 ds_write_b16 per thread and tile (136-byte pitch), plain ds_read_b64 of the fragment.
// (r03 experiment, not kept: K / V tiles by LDS-DMA into a two-stage ring per key group with the V fragments read by inline-asm
//  ds_read_b64_tr_b16 — through the builtin hipcc waits vmcnt(0), i.e. for the NEXT tile's transfer, in front of the P.V MFMAs —
//  measured 28.2 against 28.2 us with three key groups, 28.0 against 28.6 with two, and failed its parity test;
//  profiles/r03b_attn_dma_experiment.txt, commit 5eabd95.)
typedef short short4_t __attribute__((ext_vector_type(4)));
#define V_PITCH_TR 192
template <int NG, bool MASK, bool VTR>
__global__ void __launch_bounds__(256*NG) k_fattn_mfma(const FattnArgs a) {
    constexpr int TILE_LDS = KT*128 + (VTR ? KT*V_PITCH_TR : FA_D*VT_PITCH);
    constexpr int COMB_LDS = (NG - 1)*34*256*4;
    __shared__ __attribute__((aligned(16))) char lds[NG*TILE_LDS > COMB_LDS ? NG*TILE_LDS : COMB_LDS];

    const int grp = NG > 1 ? __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 8)) : 0;
    const int tid = threadIdx.x & 255, wave = tid >> 6, lane = tid & 63;
    char * ldsK = lds + grp*TILE_LDS; char * ldsV = ldsK + KT*128;
    const int hq = blockIdx.y, hk = hq / a.rk2, hv = hq / a.rv2;
    const int qi = blockIdx.x*128 + wave*32 + (lane & 31);          // this lane's query
    const int hf = lane >> 5;
    const bool q_ok = qi < a.T;
    const float c2 = a.scale * 1.4426950408889634f;                 // MASK = false: exponent scale in the exp2 domain

    // Q^T fragments (B operand): lane (query, hf) holds d = 16*kk + 8*hf + e
    half8_t qf[4];
    {
        const float * qp = (const float *) (a.q.data + (int64_t) qi*a.q.nb[1] + (int64_t) hq*a.q.nb[2]);
        #pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            float4 x0 = make_float4(0, 0, 0, 0), x1 = x0;
            if (q_ok) { x0 = *(const float4 *) (qp + 16*kk + 8*hf); x1 = *(const float4 *) (qp + 16*kk + 8*hf + 4); }
            qf[kk][0] = (half_t) x0.x; qf[kk][1] = (half_t) x0.y; qf[kk][2] = (half_t) x0.z; qf[kk][3] = (half_t) x0.w;
            qf[kk][4] = (half_t) x1.x; qf[kk][5] = (half_t) x1.y; qf[kk][6] = (half_t) x1.z; qf[kk][7] = (half_t) x1.w;
        }
    }

    floatx16 o[2];
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;

    const char * kbase = a.k.data + (int64_t) hk*a.k.nb[2];
    const char * vbase = a.v.data + (int64_t) hv*a.v.nb[2];
    const char * mrow  = MASK && a.has_mask && q_ok ? a.m.data + (int64_t) qi*a.m.nb[1] : nullptr;

    // staging: thread -> (key = tid>>3 [+32], chunk = tid&7): 16 bytes = 8 d-values
    const int skey = tid >> 3, sch = tid & 7;

    // this thread's share of a K/V tile, global -> registers.  Keys past n_kv are clamped to the last key (always a valid address:
    // an unconditional load has no basic block of its own and therefore no s_waitcnt vmcnt(0) behind it); their scores are
    // forced to -inf below, so their values never matter.
    uint4 kr0, kr1, vr0, vr1;
#define FA_FETCH(K0) do { \
        const int key0_ = min((K0) + skey, a.n_kv - 1), key1_ = min((K0) + skey + 32, a.n_kv - 1); \
        kr0 = *(const uint4 *) (kbase + (int64_t) key0_*a.k.nb[1] + sch*16); vr0 = *(const uint4 *) (vbase + (int64_t) key0_*a.v.nb[1] + sch*16); \
        kr1 = *(const uint4 *) (kbase + (int64_t) key1_*a.k.nb[1] + sch*16); vr1 = *(const uint4 *) (vbase + (int64_t) key1_*a.v.nb[1] + sch*16); \
    } while (0)
    const int nt = (a.n_kv + KT - 1) / KT, tpg = (nt + NG - 1) / NG;
    const int t_begin = grp*tpg, t_end = min(nt, t_begin + tpg);
    FA_FETCH(t_begin*KT);
    for (int it = 0; it < tpg; it++) {                            // every group makes the same number of barrier visits
        const int k0 = (t_begin + it)*KT;
        const bool live = t_begin + it < t_end;                   // uniform per group (a whole number of waves)
        __syncthreads();                                          // previous tile fully consumed
        if (live) {
            #pragma unroll
            for (int i = 0; i < 2; i++) {
                const int kl = skey + 32*i;
                const uint4 kq = i ? kr1 : kr0, vq = i ? vr1 : vr0;
                *(uint4 *) (ldsK + k_off(kl, sch)) = kq;
                if constexpr (VTR) {
                    *(uint4 *) (ldsV + kl*V_PITCH_TR + sch*16) = vq;
                } else {
                    const uint32_t w[4] = { vq.x, vq.y, vq.z, vq.w };
                    #pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const uint16_t hv16 = (uint16_t) ((w[e >> 1] >> (16*(e & 1))) & 0xFFFF);
                        *(uint16_t *) (ldsV + (sch*8 + e)*VT_PITCH + kl*2) = hv16;
                    }
                }
            }
        }
        __syncthreads();
        if (!live) continue;
        // the next tile's loads are in flight while this one is computed (clamped past the end: harmless re-read of the last keys)
        FA_FETCH(k0 + KT);

        // S^T[key][query] for the two 32-key blocks of the tile
        floatx16 s[2];
        #pragma unroll
        for (int b = 0; b < 2; b++) {
            #pragma unroll
            for (int r = 0; r < 16; r++) s[b][r] = 0.0f;
            #pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const half8_t kf = *(const half8_t *) (ldsK + k_off(b*32 + (lane & 31), kk*2 + hf));
                s[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], s[b], 0, 0, 0);
            }
        }
        // running max.  register r of block b <-> key k0 + 32b + (r&3) + 8*(r>>2) + 4*hf
        float tmax = -INFINITY;
        if (MASK) {
            #pragma unroll
            for (int b = 0; b < 2; b++) {
                #pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int key = k0 + 32*b + 8*g + 4*hf;
                    float mv[4] = { 0, 0, 0, 0 };
                    if (mrow && key < a.n_kv) {
                        if (key + 3 < a.n_kv && ((a.m.nb[1] | (uintptr_t) a.m.data) % 8 == 0)) {
                            const uint2 mm = *(const uint2 *) (mrow + key*2);
                            mv[0] = h2f((uint16_t) (mm.x & 0xFFFF)); mv[1] = h2f((uint16_t) (mm.x >> 16)); mv[2] = h2f((uint16_t) (mm.y & 0xFFFF)); mv[3] = h2f((uint16_t) (mm.y >> 16));
                        } else {
                            for (int e = 0; e < 4 && key + e < a.n_kv; e++) mv[e] = h2f(*(const uint16_t *) (mrow + (key + e)*2));
                        }
                    }
                    #pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float x = s[b][4*g + e] * a.scale + mv[e];
                        if (key + e >= a.n_kv) x = -INFINITY;
                        s[b][4*g + e] = x;
                        tmax = fmaxf(tmax, x);
                    }
                }
            }
        } else {
            if (k0 + KT > a.n_kv) {                               // the last tile only: keys past the end never count
                const int lim = a.n_kv - k0 - 4*hf;               // (hipcc if-converts this block: one compare against a constant + one select per score)
                #pragma unroll
                for (int b = 0; b < 2; b++)
                    #pragma unroll
                    for (int r = 0; r < 16; r++)
                        if (32*b + (r & 3) + 8*(r >> 2) >= lim) s[b][r] = -INFINITY;
            }
            #pragma unroll
            for (int b = 0; b < 2; b++)
                #pragma unroll
                for (int r = 0; r < 16; r++) tmax = fmaxf(tmax, s[b][r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float mc = m_new*c2;                               // the shift every p of this tile is taken against
        const float alpha = MASK ? __expf(m_run - m_new) : __builtin_amdgcn_exp2f(m_run*c2 - mc);
        m_run = m_new;
        float psum = 0.0f;
        half8_t pf[4];                                         // B operand of P^T: pf[2b + kk'][e] <-> register 8kk'+e of block b
        #pragma unroll
        for (int b = 0; b < 2; b++) {
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const float p = MASK ? __expf(s[b][r] - m_new) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[b][r], c2, -mc));
                psum += p;
                pf[2*b + (r >> 3)][r & 7] = (half_t) p;
            }
        }
        l_run = l_run*alpha + psum;
        #pragma unroll
        for (int i = 0; i < 2; i++)
            #pragma unroll
            for (int r = 0; r < 16; r++) o[i][r] *= alpha;
        // O^T[d][query] += V^T[d][key] * P^T[key][query]
        #pragma unroll
        for (int c = 0; c < 4; c++) {                          // c = 2b + kk': keys 32b + 16kk' + {(e&3) + 8(e>>2) + 4hf}
            const int kb = 16*c + 4*hf;
            #pragma unroll
            for (int i = 0; i < 2; i++) {
                half8_t vf;
                if constexpr (VTR) {
                    // this lane's 16-lane group reads the block rows (keys) 16c + 4hf + {0..3} [+ 8], columns (dims) 32i + 16(group & 1) + {0..15}
                    const int li = lane & 15;
                    const char * vb = ldsV + (16*c + 4*hf + (li >> 2))*V_PITCH_TR + (32*i + 16*((lane >> 4) & 1) + 4*(li & 3))*2;
                    const short4_t r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t *) vb);
                    const short4_t r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t *) (vb + 8*V_PITCH_TR));
                    __builtin_memcpy(&vf, &r0, 8);
                    __builtin_memcpy((char *) &vf + 8, &r1, 8);
                } else {
                    const char * vp = ldsV + (i*32 + (lane & 31))*VT_PITCH + kb*2;
                    const uint2 v0 = *(const uint2 *) vp, v1 = *(const uint2 *) (vp + 16);
                    const uint32_t vw[4] = { v0.x, v0.y, v1.x, v1.y };
                    __builtin_memcpy(&vf, vw, 16);
                }
                o[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[c], o[i], 0, 0, 0);
            }
        }
    }
#undef FA_FETCH

    if (NG > 1) {
        // merge the key groups' records: comb[g-1][field][tid], field 0 = m, 1 = l, 2 + 16i + r = o[i][r] (the tile buffers are dead)
        float * comb = (float *) lds;
        __syncthreads();
        if (grp > 0) {
            float * rec = comb + (grp - 1)*34*256 + tid;
            rec[0] = m_run; rec[256] = l_run;
            #pragma unroll
            for (int i = 0; i < 2; i++)
                #pragma unroll
                for (int r = 0; r < 16; r++) rec[(2 + 16*i + r)*256] = o[i][r];
        }
        __syncthreads();
        if (grp > 0) return;
        #pragma unroll
        for (int g = 1; g < NG; g++) {
            const float * rec = comb + (g - 1)*34*256 + tid;
            const float m_g = rec[0], l_g = rec[256];
            const float m_new = fmaxf(m_run, m_g);
            const float a0 = MASK ? __expf(m_run - m_new) : __builtin_amdgcn_exp2f(m_run*c2 - m_new*c2);
            const float a1 = MASK ? __expf(m_g - m_new)   : __builtin_amdgcn_exp2f(m_g*c2 - m_new*c2);
            m_run = m_new;
            l_run = l_run*a0 + l_g*a1;
            #pragma unroll
            for (int i = 0; i < 2; i++)
                #pragma unroll
                for (int r = 0; r < 16; r++) o[i][r] = o[i][r]*a0 + rec[(2 + 16*i + r)*256]*a1;
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot == 0.0f ? 0.0f : 1.0f / l_tot;
    if (q_ok) {
        float * dp = (float *) (a.d.data + (int64_t) hq*a.d.nb[1] + (int64_t) qi*a.d.nb[2]);
        #pragma unroll
        for (int i = 0; i < 2; i++)
            #pragma unroll
            for (int g = 0; g < 4; g++) {
                const int d0 = 32*i + 8*g + 4*hf;
                *(float4 *) (dp + d0) = make_float4(o[i][4*g]*inv, o[i][4*g+1]*inv, o[i][4*g+2]*inv, o[i][4*g+3]*inv);
            }
    }
    if (a.rq) {
        // Q8_0 rows of the result seen as [T][H*64] (quantize_row_q8_0, arch/x86/quants.c:302-398): a block = 32 consecutive dims of one
        // head = the 16 registers o[i][*] of this lane and of lane^32 (same query)
        #pragma unroll
        for (int i = 0; i < 2; i++) {
            float v[16], amax = 0.0f;
            #pragma unroll
            for (int r = 0; r < 16; r++) { v[r] = o[i][r]*inv; amax = fmaxf(amax, fabsf(v[r])); }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            const float d  = amax / 127.0f;
            const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
            if (q_ok) {
                int8_t * pp = a.rq + (int64_t) qi*a.prep_ld + hq*FA_D;
                #pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int q0 = (int) rintf(v[4*g]*id), q1 = (int) rintf(v[4*g+1]*id), q2 = (int) rintf(v[4*g+2]*id), q3 = (int) rintf(v[4*g+3]*id);
                    *(uint32_t *) (pp + 32*i + 8*g + 4*hf) = (uint32_t) (q0 & 0xFF) | ((uint32_t) (q1 & 0xFF) << 8) | ((uint32_t) (q2 & 0xFF) << 16) | ((uint32_t) (q3 & 0xFF) << 24);
                }
                if (hf == 0) a.rd[(int64_t) (hq*2 + i)*a.T + qi] = round_f16(d);
            }
        }
    } else if (a.prep) {
        // The result, seen as [T][H*64], is the activation matrix of the output projection (src/whisper.cpp:2165-2167 -> :2199-2203):
        // leave what k_prep_act (mode 1) would make of it.  A Q8_0 block = 32 consecutive dims of one head = the 16 registers
        // o[i][*] of this lane and of lane^32 (same query).
        #pragma unroll
        for (int i = 0; i < 2; i++) {
            float v[16], amax = 0.0f;
            #pragma unroll
            for (int r = 0; r < 16; r++) { v[r] = o[i][r]*inv; amax = fmaxf(amax, fabsf(v[r])); }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            const float d  = round_f16(amax / 127.0f);
            const float id = amax != 0.0f ? 127.0f / amax : 0.0f;
            if (q_ok) {
                uint16_t * pp = a.prep + (int64_t) qi*a.prep_ld + hq*FA_D;
                #pragma unroll
                for (int g = 0; g < 4; g++) {
                    float q[4];
                    #pragma unroll
                    for (int e = 0; e < 4; e++) q[e] = fminf(fmaxf(d * rintf(v[4*g + e]*id), -65504.0f), 65504.0f);
                    *(uint2 *) (pp + 32*i + 8*g + 4*hf) = make_uint2(f2h(q[0]) | ((uint32_t) f2h(q[1]) << 16), f2h(q[2]) | ((uint32_t) f2h(q[3]) << 16));
                }
            }
        }
    }
}

static int flash_attn_impl(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v,
                           const mi355x_tensor * mask, const mi355x_tensor * dst, float scale, void * prep_out);

extern "C" int mi355x_flash_attn_ext(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v,
                                     const mi355x_tensor * mask, const mi355x_tensor * dst, float scale) {
    return flash_attn_impl(ctx, q, k, v, mask, dst, scale, nullptr);
}

// T > 8 only: the same, also leaving mi355x_prep_act(mode 1) of dst viewed as [T][H*64] in prep_out (f16): the activations of the
// quantized-weight output projection that follows.  dst must be contiguous in that view.
extern "C" int mi355x_flash_attn_ext_prep(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v,
                                          const mi355x_tensor * mask, const mi355x_tensor * dst, float scale, void * prep_out) {
    if (!prep_out || ((uintptr_t) prep_out % 16) || q->ne[1] <= 8 || k->ne[1] == 0) return MI355X_E_UNSUPPORTED;
    if (dst->nb[1] != FA_D*4 || dst->nb[2] != dst->ne[1]*FA_D*4) return MI355X_E_UNSUPPORTED;
    return flash_attn_impl(ctx, q, k, v, mask, dst, scale, prep_out);
}

extern "C" int mi355x_flash_attn_ext_prep_rows(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v,
                                               const mi355x_tensor * mask, const mi355x_tensor * dst, float scale, void * rows_out) {
    if (!rows_out || ((uintptr_t) rows_out % 16) || q->ne[1] <= 8 || k->ne[1] == 0 || (q->ne[2] * FA_D) % 128) return MI355X_E_UNSUPPORTED;
    if (dst->nb[1] != FA_D*4 || dst->nb[2] != dst->ne[1]*FA_D*4) return MI355X_E_UNSUPPORTED;
    return flash_attn_impl(ctx, q, k, v, mask, dst, scale, (void *) ((uintptr_t) rows_out | 1));      // bit 0: rows, not f16 (pointers are 16-byte aligned)
}

static int flash_attn_impl(mi355x_ctx * ctx, const mi355x_tensor * q, const mi355x_tensor * k, const mi355x_tensor * v,
                           const mi355x_tensor * mask, const mi355x_tensor * dst, float scale, void * prep_out) {
    if (q->type != MI355X_TYPE_F32 || k->type != MI355X_TYPE_F16 || v->type != MI355X_TYPE_F16 || dst->type != MI355X_TYPE_F32) return MI355X_E_UNSUPPORTED;
    if (q->ne[0] != FA_D || k->ne[0] != FA_D || v->ne[0] != FA_D || dst->ne[0] != FA_D) return MI355X_E_UNSUPPORTED;
    if (q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return MI355X_E_UNSUPPORTED;
    if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2 || dst->nb[0] != 4) return MI355X_E_UNSUPPORTED;
    const int T = (int) q->ne[1], H = (int) q->ne[2], n_kv = (int) k->ne[1];
    if (v->ne[1] != n_kv || dst->ne[1] != H || dst->ne[2] != T) return MI355X_E_UNSUPPORTED;
    if (k->ne[2] <= 0 || H % k->ne[2] || v->ne[2] <= 0 || H % v->ne[2]) return MI355X_E_UNSUPPORTED;
    // 16-byte vector accesses
    if (((uintptr_t) q->data | q->nb[1] | q->nb[2]) % 16 || ((uintptr_t) k->data | k->nb[1] | k->nb[2]) % 16 ||
        ((uintptr_t) v->data | v->nb[1] | v->nb[2]) % 16 || ((uintptr_t) dst->data | dst->nb[1] | dst->nb[2]) % 16) return MI355X_E_UNSUPPORTED;
    if (mask && (mask->type != MI355X_TYPE_F16 || mask->ne[0] < n_kv || mask->ne[1] < T || mask->nb[0] != 2 || mask->ne[2] != 1 || mask->ne[3] != 1)) return MI355X_E_UNSUPPORTED;
    if (T == 0 || H == 0) return 0;

    FattnArgs a; memset(&a, 0, sizeof(a));
    a.q = to_d(q); a.k = to_d(k); a.v = to_d(v); a.d = to_d(dst);
    if (mask) a.m = to_d(mask);
    a.has_mask = mask != nullptr; a.scale = scale; a.T = T; a.n_kv = n_kv; a.H = H;
    a.rk2 = (int) (H / k->ne[2]); a.rv2 = (int) (H / v->ne[2]);
    a.prep_ld = H * FA_D;
    if ((uintptr_t) prep_out & 1) {
        const qrows_t R = qrows_of((void *) ((uintptr_t) prep_out & ~(uintptr_t) 1), 0, (int64_t) H * FA_D, T);
        a.rq = R.q; a.rd = R.d;
    } else a.prep = (uint16_t *) prep_out;
    const double kv_bytes = 2.0 * n_kv * FA_D * 2 * H;
    const double flops = 4.0 * T * (double) n_kv * FA_D * H;
    if (n_kv == 0) return mi355x_memset(ctx, dst->data, 0, (size_t) dst->nb[3]*dst->ne[3]);

    if (T <= 8) {               // decoder step: partial records per 128-key chunk (decode.hip), merged here when no projection consumes them
        mi355x_attn_partials parts;
        const int rc = mi355x_flash_attn_partial(ctx, q, k, v, mask, scale, &parts);
        if (rc == 0) return mi355x_flash_attn_combine(ctx, &parts, dst);
        if (rc != MI355X_E_UNSUPPORTED) return rc;
    }
    const double bytes = kv_bytes * ((T + 127) / 128) + (double) T*H*FA_D*8;
    // key groups per workgroup (waves per SIMD): 3 from 12 key tiles on, 2 from 4 on (test option MI355X_OPT_FATTN_NG = 1..4 forces one).
    // Encoder size, 1500 x 1500 x 20 heads, hipEvent means (profiles/r03b_encoder_kernel_variants.txt): 1 group 40.2 us (r02's kernel: 49),
    // 2 groups 33.6, 3 groups 32.4, 4 groups 33.3
    const int ng_env = mi355x_opt(MI355X_OPT_FATTN_NG, 0);
    const int ntiles = (n_kv + KT - 1) / KT;
    const bool folded = !mask && scale > 0.0f;                  // the running maximum on raw scores needs a positive scale
    int ng = ng_env >= 1 && ng_env <= 4 ? ng_env : (ntiles >= 12 ? 3 : ntiles >= 4 ? 2 : 1);
    if (ng > ntiles) ng = ntiles;
    if (!folded && ng > 3) ng = 3;                              // the masked form needs 166 VGPRs: three waves per SIMD at most
    const dim3 grid((T + 127) / 128, H);
    // V goes through the transposing LDS read (ds_read_b64_tr_b16); round 2's layout — transposed on the way into LDS — is gone
#define FA_LAUNCH(NG_) (folded ? emit(ctx, "fattn_mfma", k_fattn_mfma<NG_, false, true>, grid, dim3(256*NG_), 0, a, bytes, flops) \
                               : emit(ctx, "fattn_mfma", k_fattn_mfma<NG_, true, true>,  grid, dim3(256*NG_), 0, a, bytes, flops))
    switch (ng) {
        case 4:  return emit(ctx, "fattn_mfma", k_fattn_mfma<4, false, true>, grid, dim3(1024), 0, a, bytes, flops);
        case 3:  return FA_LAUNCH(3);
        case 2:  return FA_LAUNCH(2);
        default: return FA_LAUNCH(1);
    }
#undef FA_LAUNCH
}
