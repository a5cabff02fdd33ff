// mi_planner.cpp — part of the MI355X ggml backend plugin; see mi_backend.h for the map of the translation units.
#include "mi_backend.h"

mi355x_tensor to_mt(const ggml_tensor * t) {
    mi355x_tensor m;
    m.data = t->data; m.type = (int32_t) t->type; m.reserved = 0;
    for (int i = 0; i < 4; i++) { m.ne[i] = t->ne[i]; m.nb[i] = (int64_t) t->nb[i]; }
    return m;
}

bool op_is_empty(const ggml_tensor * t) {
    return t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE;
}

int use_count(const ggml_cgraph * g, const ggml_tensor * t) {
    if (!g->use_counts || !g->visited_hash_set.keys) return 1 << 20;
    const size_t pos = ggml_hash_find(&g->visited_hash_set, t);
    if (pos == GGML_HASHSET_FULL || !ggml_bitset_get(g->visited_hash_set.used, pos)) return 1 << 20;
    return g->use_counts[pos];
}
// may `t` be elided (computed only inside a fused kernel) given that exactly `n` fused consumers read it?
bool can_elide(const ggml_cgraph * g, const ggml_tensor * t, int n) {
    return use_count(g, t) == n && !t->view_src && !(t->flags & GGML_TENSOR_FLAG_OUTPUT);
}

bool overlap(const void * a, size_t na, const void * b, size_t nb) {
    const char * pa = (const char *) a, * pb = (const char *) b;
    return pa < pb + nb && pb < pa + na;
}
bool t_overlap(const ggml_tensor * a, const ggml_tensor * b) { return overlap(a->data, ggml_nbytes(a), b->data, ggml_nbytes(b)); }

bool is_vec_f32(const ggml_tensor * t, int64_t n) {     // contiguous f32 vector of n elements (any trailing 1-dims)
    return t->type == GGML_TYPE_F32 && ggml_nelements(t) == n && t->ne[0] == n && t->nb[0] == 4;
}

// next node index after `i` that is not an empty op (or n_nodes)
int next_real(const ggml_cgraph * g, int i) {
    int j = i + 1;
    while (j < g->n_nodes && (op_is_empty(g->nodes[j]) || !(g->nodes[j]->flags & GGML_TENSOR_FLAG_COMPUTE))) j++;
    return j;
}


bool parse_mm_chain(const ggml_cgraph * g, int i, bool fuse, mm_chain & c) {
    const ggml_tensor * mm = g->nodes[i];
    if (mm->op != GGML_OP_MUL_MAT) return false;
    c = mm_chain(); c.mm = mm; c.last = mm; c.end = i;
    if (!fuse) return true;
    const ggml_tensor * w = mm->src[0], * x = mm->src[1];
    if (ggml_n_dims(w) > 2 || x->ne[2] != 1 || x->ne[3] != 1 || mm->type != GGML_TYPE_F32) return true;
    const int64_t N = mm->ne[0];
    const ggml_tensor * cur = mm;
    int stage = 0;   // 0: bias allowed, 1: scale, 2: gelu, 3: residual, 4: cpy
    int j = next_real(g, i);
    // ggml_conv_1d (ggml.c:4537-4554) puts a RESHAPE between its mul_mat and the bias add (src/whisper.cpp:2013-2020): the add reads the
    // product through a same-shape contiguous view, and its bias has one value per COLUMN of the product ([1, OC] against [OL, OC])
    if (j < g->n_nodes && g->nodes[j]->op == GGML_OP_ADD && can_elide(g, mm, 1) && mm->ne[2] == 1 && mm->ne[3] == 1) {
        const ggml_tensor * n = g->nodes[j];
        for (int sl = 0; sl < 2; sl++) {
            const ggml_tensor * v = n->src[sl], * o = n->src[1 - sl];
            if (v->op == GGML_OP_RESHAPE && v->src[0] == mm && v->data == mm->data && ggml_are_same_shape(v, mm) && ggml_is_contiguous(v) &&
                use_count(g, v) == 1 && !(v->flags & GGML_TENSOR_FLAG_OUTPUT) && ggml_are_same_shape(n, mm) && n->type == GGML_TYPE_F32 &&
                o->type == GGML_TYPE_F32 && o->ne[0] == 1 && o->ne[1] == mm->ne[1] && ggml_nelements(o) == mm->ne[1] && o->nb[1] == 4 && mm->ne[1] > 8) {
                c.ep.bias = (const float *) o->data; c.ep.bias_per_col = 1;
                stage = 1; cur = n; c.last = n; c.end = j;
                j = next_real(g, j);
                break;
            }
        }
    }
    while (j < g->n_nodes && stage < 5) {
        const ggml_tensor * n = g->nodes[j];
        if (!can_elide(g, cur, 1)) break;
        bool took = false;
        if (n->op == GGML_OP_ADD && (n->src[0] == cur || n->src[1] == cur) && ggml_are_same_shape(n, cur) && n->type == GGML_TYPE_F32) {
            const ggml_tensor * o = n->src[0] == cur ? n->src[1] : n->src[0];
            if (stage <= 0 && is_vec_f32(o, N) && o != cur) { c.ep.bias = (const float *) o->data; stage = 1; took = true; }
            else if (stage <= 3 && o->type == GGML_TYPE_F32 && ggml_are_same_shape(o, cur) && o->nb[0] == 4 && o != cur) {
                c.ep.residual = (const float *) o->data; c.ep.residual_nb1 = (int64_t) o->nb[1]; stage = 4; took = true;
                c.res_node = j; c.res_slot = n->src[0] == cur ? 1 : 0;
            }
        } else if (n->op == GGML_OP_SCALE && n->src[0] == cur && stage <= 1 && ggml_get_op_params_f32(n, 1) == 0.0f) {
            c.ep.scale = ggml_get_op_params_f32(n, 0); c.ep.has_scale = 1; stage = 2; took = true;
        } else if (n->op == GGML_OP_UNARY && ggml_get_unary_op(n) == GGML_UNARY_OP_GELU && n->src[0] == cur && stage <= 2) {
            c.ep.gelu = 1; stage = 3; took = true;
        } else if (n->op == GGML_OP_CPY && n->src[0] == cur && n->type == GGML_TYPE_F16 && ggml_is_contiguous(n) &&
                   ggml_nelements(n) == ggml_nelements(cur) && ggml_is_contiguous(cur)) {
            stage = 5; took = true;
        }
        if (!took) break;
        cur = n; c.last = n; c.end = j;
        j = next_real(g, j);
    }
    // memory hazards: the result must not land on anything the kernel still reads
    const ggml_tensor * res_t = nullptr;
    if (c.last != mm) {
        bool bad = t_overlap(c.last, x) || t_overlap(c.last, w);
        if (c.ep.residual) {
            // identical aliasing (in-place add) is fine: every element is read before it is written by the same lane
            const char * r = (const char *) c.ep.residual;
            const size_t rn = (size_t) c.ep.residual_nb1 * (size_t) mm->ne[1];
            if (overlap(c.last->data, ggml_nbytes(c.last), r, rn) && !(r == (const char *) c.last->data && c.ep.residual_nb1 == (int64_t) c.last->nb[1])) bad = true;
        }
        if (c.ep.bias && overlap(c.last->data, ggml_nbytes(c.last), c.ep.bias, (c.ep.bias_per_col ? mm->ne[1] : N)*4)) bad = true;
        (void) res_t;
        if (bad) { c = mm_chain(); c.mm = mm; c.last = mm; c.end = i; }
    }
    return true;
}

int mode_for(ggml_type t) { return t == GGML_TYPE_Q4_K ? 2 : (is_quant_type(t) ? 1 : 0); }
// Which GEMM family takes a quantized weight x more than 8 columns (GGML_MI355X_MMQ, default 1):
//   1  by width.  T >= MI_WIDE_MIN_T columns (the encoder and the cross-K/V products: 1500): f16 MFMA on a one-time f16 copy of the weight through the LDS-DMA
//      ring (gemm_mfma.hip: k_gemm_f16_ring), f16(d*q) activations that keep the reference's Q8_0 / Q8_K rounding decisions.  Narrower products (prompt, beam /
//      batch steps beyond the mat-vec range): the int8 tile GEMM over the quantized operands (mmq.hip: the CPU's own integer sums, activations = the reference's
//      blocks as integers: "rows", prep modes 3 / 4).  Why by width: the int8 GEMM's per-block fix-up is VALU work that costs the encoder 1.0 (Q5_0) .. 2.5 ms
//      (Q4_K) per chunk for a kernel-level 1e-10 that buys nothing at model level (profiles/r05_regress_ab.txt), and costs nothing at the
//      narrow widths; the f16 copies are 2 bytes per encoder / cross-attention weight (1.4 GB of 288 for large-v3), made on first use.
//   2  the int8 tile GEMM at every width (rounds 4-5)        0  the f16 ring at every width (rounds 2-3)
//   3  like 1, but the wide products on k_gemm_dq: the weight's quantized planes unpacked per workgroup into LDS, no f16 copy (round 6; bit-identical to the ring
//      on the copy, measured 10-150 % slower: profiles/r06_gemm_dq_anatomy.txt — the vector-memory path serves ~0.2 cache lines per clock and CU, and the
//      row-scattered block planes cost more line requests than the copy's contiguous rows)
int mi_mmq_mode() { static const int m = [] { const char * e = getenv("GGML_MI355X_MMQ"); return e ? std::max(0, std::min(3, atoi(e))) : 1; }(); return m; }
bool mi_mmq_on() { return mi_mmq_mode() != 0; }
#define MI_WIDE_MIN_T 1024
static bool mi_wide_f16(int64_t T) { const int m = mi_mmq_mode(); return m == 0 || ((m == 1 || m == 3) && T >= MI_WIDE_MIN_T); }
int rows_mode_for(const ggml_tensor * w, int64_t K, int64_t T) {
    if (mi_wide_f16(T) || !is_quant_type(w->type) || K % 128 != 0) return 0;
    return w->type == GGML_TYPE_Q4_K ? (K % 256 == 0 ? 4 : 0) : 3;
}

// f16 copy of a quantized weight for the MFMA path (nullptr: not eligible / over budget -> the GEMM dequantizes in its loop)
const void * mi_shadow_get(mi_backend_ctx * b, const ggml_tensor * w, const mi355x_tensor & mw) {
    constexpr size_t cap_mb = 16384;               // f16 copies of quantized weights (GGML_MI355X_MMQ=0 only): at most 16 GB of the 288
    if (cap_mb == 0) return nullptr;
    ggml_backend_buffer_t buf = w->view_src ? w->view_src->buffer : w->buffer;
    if (!buf || !mi_buffer_is_ours(buf) || buf->usage != GGML_BACKEND_BUFFER_USAGE_WEIGHTS) return nullptr;
    std::lock_guard<std::mutex> lk(g_shadow_mtx);
    auto it = g_shadows.find(w->data);
    if (it != g_shadows.end()) {
        const mi_shadow & sh = it->second;
        return (sh.type == (int) w->type && sh.ne0 == w->ne[0] && sh.ne1 == w->ne[1]) ? sh.f16 : nullptr;
    }
    const size_t bytes = (size_t) w->ne[0] * (size_t) w->ne[1] * 2;
    if (g_shadow_bytes + bytes > cap_mb * 1024 * 1024) return nullptr;
    const mi_buffer_ctx * bc = (const mi_buffer_ctx *) buf->context;
    void * p = nullptr;
    if (hipMalloc(&p, bytes + 256) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    // the copy is published only once it is complete, so another backend (another whisper_state on its own stream) never reads it early
    if (mi355x_dequant_f16(b->k, &mw, p) != 0 || mi355x_ctx_synchronize(b->k) != 0) { (void) hipFree(p); return nullptr; }
    g_shadows[w->data] = { p, bytes, bc->base, bc->device, (int) w->type, w->ne[0], w->ne[1] };
    g_shadow_bytes += bytes;
    g_shadow_count.store(g_shadows.size());
    return p;
}

// room for T x K prepared f16 activations in the backend's scratch; 0 ok, > 0 error
int mi_act_reserve(mi_backend_ctx * b, size_t need, bool alt) {
    void * & buf = alt ? b->act_alt : b->act;
    size_t & size = alt ? b->act_alt_size : b->act_size;
    if (need <= size) return 0;
    mi355x_ctx_synchronize(b->k);                 // (also sends held-back launches that still name the old buffer)
    if (buf) (void) hipFree(buf);
    buf = nullptr; size = 0;
    if (!alt) b->act_src = nullptr;
    const size_t sz = need + (need >> 2);
    if (hipMalloc(&buf, sz) != hipSuccess) return (int) hipErrorOutOfMemory;
    size = sz;
    return 0;
}

// does tensor x (F32 [K, T]) feed the MFMA GEMM path as the activation of mul_mat `mm`?  (the conditions of run_mm_chain)
bool mm_takes_prepared(const mi_backend_ctx * b, const ggml_tensor * mm, const ggml_tensor * x, int & mode_out) {
    if (mm->op != GGML_OP_MUL_MAT || mm->src[1] != x || b->exact) return false;
    const ggml_tensor * w = mm->src[0];
    const int64_t K = w->ne[0], T = x->ne[1];
    const bool two_d = ggml_n_dims(w) <= 2 && x->ne[2] == 1 && x->ne[3] == 1;
    if (!(two_d && T > 8 && K % 8 == 0 && x->type == GGML_TYPE_F32 && x->nb[0] == 4 && (x->nb[1] % 16 == 0) && ((uintptr_t) x->data % 16 == 0))) return false;
    if (!((is_quant_type(w->type) && ggml_is_contiguous(w)) || (w->type == GGML_TYPE_F16 && w->nb[0] == 2 && w->nb[1] % 16 == 0))) return false;
    const int mode = mode_for(w->type);
    if ((mode == 1 && K % 32) || (mode == 2 && K % 256)) return false;
    int rmode = rows_mode_for(w, K, T);
    if (rmode && ((uintptr_t) w->data % 16)) rmode = 0;                   // (mi355x_gemm_q8act's own precondition: never promise rows it would refuse)
    mode_out = rmode ? rmode : mode;
    return true;
}

int run_mm_chain(mi_backend_ctx * b, const mm_chain & c, const ggml_cgraph * g) {
    const ggml_tensor * mm = c.mm, * w = mm->src[0], * x = mm->src[1];
    // a producer elided this F32 activation because mm_takes_prepared() promised that its prepared rows would be consumed.  If the
    // consuming kernel then refuses them (alignment, a shape only its launch code knows), no path that re-reads x->data may run: fail loudly.
    auto reads_elided = [&]() {
        if (!b->elided_src || mm != b->elided_for || x->data != b->elided_src) return false;
        GGML_LOG_ERROR("ggml-mi355x: %s: the prepared activations of %s were refused and its F32 form was never stored\n", mm->name, x->name);
        return true;
    };
    mi355x_tensor mw = to_mt(w), mx = to_mt(x);
    // destination: the chain's last tensor, seen as [N, T] with the dtype of that tensor
    mi355x_tensor md = to_mt(mm);
    md.data = c.last->data; md.type = (int32_t) c.last->type;
    if (c.last->type == GGML_TYPE_F16) { md.nb[0] = 2; md.nb[1] = mm->ne[0]*2; md.nb[2] = md.nb[1]*mm->ne[1]; md.nb[3] = md.nb[2]; }
    else if (c.last != mm)              { md.nb[0] = 4; md.nb[1] = (int64_t) c.last->nb[1]; md.nb[2] = (int64_t) c.last->nb[2]; md.nb[3] = (int64_t) c.last->nb[3]; }
    const bool has_ep = c.ep.bias || c.ep.has_scale || c.ep.gelu || c.ep.residual;
    const int64_t K = w->ne[0], T = x->ne[1];
    const bool two_d = ggml_n_dims(w) <= 2 && x->ne[2] == 1 && x->ne[3] == 1;

    if (b->exact && two_d && T > 8 && is_quant_type(w->type) && x->type == GGML_TYPE_F32) {
        // reference-exact mode: the integer-dot mat-vec kernels on 8-column slices (same integer sums as the CPU's vec_dot, f32
        // scale-accumulate) instead of the MFMA path whose activations are f16-rounded d*q products
        for (int64_t t0 = 0; t0 < T; t0 += 8) {
            const int64_t nt = std::min<int64_t>(8, T - t0);
            mi355x_tensor sx = mx, sd = md;
            sx.data = (char *) mx.data + t0 * mx.nb[1]; sx.ne[1] = nt;
            sd.data = (char *) md.data + t0 * md.nb[1]; sd.ne[1] = nt;
            mi355x_epilogue ep = c.ep;
            if (ep.residual) ep.residual = (const float *) ((const char *) ep.residual + t0 * ep.residual_nb1);
            if (ep.bias && ep.bias_per_col) ep.bias += t0;
            const int rc = mi355x_mul_mat(b->k, &mw, &sx, &sd, has_ep ? &ep : nullptr);
            if (rc) return rc;
        }
        return 0;
    }

    // MFMA path with shared prepared activation
    if (two_d && T > 8 && K % 8 == 0 && x->nb[0] == ggml_type_size(x->type) && (x->nb[1] % 16 == 0) && ((uintptr_t) x->data % 16 == 0) &&
        (x->type == GGML_TYPE_F32 || x->type == GGML_TYPE_F16) &&
        ((is_quant_type(w->type) && ggml_is_contiguous(w)) || (w->type == GGML_TYPE_F16 && w->nb[0] == 2 && w->nb[1] % 16 == 0))) {
        const int mode = mode_for(w->type);
        const int rmode = rows_mode_for(w, K, T);
        if (rmode && ggml_is_contiguous(w)) {
            // the int8 tile GEMM on the quantized weight and the activation rows
            const int rr = mi_act_reserve(b, (size_t) T * K * 2);          // (rows need 1.25 bytes per element: the f16 size covers them)
            if (rr != 0) return rr;
            if (!(b->act_src == x->data && b->act_K == K && b->act_T == T && b->act_mode == rmode && b->act_nb1 == (int64_t) x->nb[1])) {
                if (reads_elided()) return (int) hipErrorInvalidValue;
                const int rc = mi355x_prep_act(b->k, x->data, (int64_t) x->nb[1], x->type == GGML_TYPE_F16, b->act, (int) K, T, rmode);
                if (rc && rc != MI355X_E_UNSUPPORTED) return rc;
                if (rc == 0) { b->act_src = x->data; b->act_K = K; b->act_T = T; b->act_mode = rmode; b->act_nb1 = (int64_t) x->nb[1]; }
                else b->act_src = nullptr;
            }
            if (b->act_src == x->data && b->act_mode == rmode) {
                // fc1 + GELU -> fc2: the epilogue writes the next product's rows (second scratch), and the F32 result only if somebody reads it
                const int64_t M = mm->ne[0];
                int rc = MI355X_E_UNSUPPORTED;
                if (g && b->fuse && c.last->type == GGML_TYPE_F32 && M % 128 == 0 && M <= 8192 &&
                    c.last->nb[0] == 4 && (int64_t) c.last->nb[1] == M*4 && c.last->ne[1] == T && c.last->ne[2] == 1 && c.last->ne[3] == 1) {
                    const int j = next_real(g, c.end);
                    int mode2 = -1;
                    if (j < g->n_nodes && mm_takes_prepared(b, g->nodes[j], c.last, mode2) && mode2 == 3 && mi_act_reserve(b, (size_t) T * M * 2, true) == 0) {
                        const bool only = can_elide(g, c.last, 1);
                        rc = mi355x_gemm_q8act_prep(b->k, &mw, b->act, T, only ? nullptr : md.data, md.nb[1], has_ep ? &c.ep : nullptr, b->act_alt);
                        if (rc == 0) {
                            std::swap(b->act, b->act_alt); std::swap(b->act_size, b->act_alt_size);
                            b->act_src = c.last->data; b->act_K = M; b->act_T = T; b->act_mode = 3; b->act_nb1 = M*4;
                            if (only) { b->elided_src = c.last->data; b->elided_for = g->nodes[j]; }
                            return 0;
                        }
                        if (rc != MI355X_E_UNSUPPORTED) return rc;
                    }
                }
                rc = mi355x_gemm_q8act(b->k, &mw, b->act, T, md.data, md.nb[1], md.type, has_ep ? &c.ep : nullptr);
                if (rc != MI355X_E_UNSUPPORTED) return rc;
            }
        }
        if (!((mode == 1 && K % 32) || (mode == 2 && K % 256))) {
            const void * act; int64_t ld;
            if (x->type == GGML_TYPE_F16 && mode == 0) { act = x->data; ld = (int64_t) x->nb[1] / 2; }
            else {
                const int rr = mi_act_reserve(b, (size_t) T * K * 2);
                if (rr != 0) return rr;
                if (!(b->act_src == x->data && b->act_K == K && b->act_T == T && b->act_mode == mode && b->act_nb1 == (int64_t) x->nb[1])) {
                    if (reads_elided()) return (int) hipErrorInvalidValue;        // (the prepared form on hand is not this path's: x->data would be read)
                    const int rc = mi355x_prep_act(b->k, x->data, (int64_t) x->nb[1], x->type == GGML_TYPE_F16, b->act, (int) K, T, mode);
                    if (rc) return rc;
                    b->act_src = x->data; b->act_K = K; b->act_T = T; b->act_mode = mode; b->act_nb1 = (int64_t) x->nb[1];
                }
                act = b->act; ld = K;
            }
            // Is the result itself the activation matrix of the next node's MFMA GEMM (fc1 + GELU -> fc2)?  Then the epilogue writes
            // that GEMM's prepared f16 activations (second scratch; this product still reads the first), and when nothing else reads
            // the F32 result it is not stored at all: one launch, a 30 MB write and a 30 MB read less per encoder layer of large-v3.
            void * prep_out = nullptr; bool prep_only = false; const ggml_tensor * prep_for = nullptr;
            const int64_t M = mm->ne[0];
            if (g && b->fuse && act == b->act && c.last->type == GGML_TYPE_F32 && M % 32 == 0 && M <= 8192 &&
                c.last->nb[0] == 4 && (int64_t) c.last->nb[1] == M*4 && c.last->ne[1] == T && c.last->ne[2] == 1 && c.last->ne[3] == 1) {
                const int j = next_real(g, c.end);
                int mode2 = -1;
                if (j < g->n_nodes && mm_takes_prepared(b, g->nodes[j], c.last, mode2) && mode2 == 1 && mi_act_reserve(b, (size_t) T * M * 2, true) == 0) {
                    prep_out = b->act_alt; prep_for = g->nodes[j];
                    prep_only = can_elide(g, c.last, 1);
                }
            }
            auto gemm = [&](const mi355x_tensor & wt) {
                if (prep_out) {
                    const int rc = mi355x_gemm_f16act_prep(b->k, &wt, act, ld, T, prep_only ? nullptr : md.data, md.nb[1], has_ep ? &c.ep : nullptr, prep_out);
                    if (rc == 0) {       // the next GEMM finds its activations prepared: the scratch buffers trade places
                        std::swap(b->act, b->act_alt); std::swap(b->act_size, b->act_alt_size);
                        b->act_src = c.last->data; b->act_K = M; b->act_T = T; b->act_mode = 1; b->act_nb1 = M*4;
                        if (prep_only) { b->elided_src = c.last->data; b->elided_for = prep_for; }
                        return 0;
                    }
                    if (rc != MI355X_E_UNSUPPORTED) return rc;
                }
                return mi355x_gemm_f16act(b->k, &wt, act, ld, T, md.data, md.nb[1], md.type, has_ep ? &c.ep : nullptr);
            };
            // wide activations: run the GEMM on the weight's f16 copy (same values, no dequantization in the loop)
            constexpr int shadow_min_t = 128;
            if (mode != 0 && T >= shadow_min_t && mi_wide_f16(T) && mi_mmq_mode() != 3) {
                if (const void * f16 = mi_shadow_get(b, w, mw)) {
                    mi355x_tensor ms = mw;
                    ms.data = (void *) f16; ms.type = MI355X_TYPE_F16;
                    ms.nb[0] = 2; ms.nb[1] = K*2; ms.nb[2] = ms.nb[1]*w->ne[1]; ms.nb[3] = ms.nb[2];
                    const int rc = gemm(ms);
                    if (rc != MI355X_E_UNSUPPORTED) return rc;
                }
            }
            const int rc = gemm(mw);
            if (rc != MI355X_E_UNSUPPORTED) return rc;
        }
    }
    if (reads_elided()) return (int) hipErrorInvalidValue;                // the generic kernel reads x->data
    return mi355x_mul_mat(b->k, &mw, &mx, &md, has_ep ? &c.ep : nullptr);
}


void parse_ln_chain(const ggml_cgraph * g, int i, bool fuse, ln_chain & c) {
    const ggml_tensor * nrm = g->nodes[i];
    c = ln_chain(); c.norm = nrm; c.last = nrm; c.end = i;
    if (!fuse || nrm->type != GGML_TYPE_F32) return;
    const int64_t n = nrm->ne[0];
    int j = next_real(g, i);
    if (j >= g->n_nodes || !can_elide(g, nrm, 1)) return;
    const ggml_tensor * m = g->nodes[j];
    if (m->op != GGML_OP_MUL || !ggml_are_same_shape(m, nrm)) return;
    const ggml_tensor * wv = m->src[0] == nrm ? m->src[1] : (m->src[1] == nrm ? m->src[0] : nullptr);
    if (!wv || !is_vec_f32(wv, n)) return;
    int j2 = next_real(g, j);
    if (j2 >= g->n_nodes || !can_elide(g, m, 1)) return;
    const ggml_tensor * a = g->nodes[j2];
    if (a->op != GGML_OP_ADD || !ggml_are_same_shape(a, m)) return;
    const ggml_tensor * bv = a->src[0] == m ? a->src[1] : (a->src[1] == m ? a->src[0] : nullptr);
    if (!bv || !is_vec_f32(bv, n)) return;
    const ggml_tensor * x = nrm->src[0];
    // result memory may alias x exactly (row-wise in-place), but must not partially overlap it
    if (t_overlap(a, x) && !(a->data == x->data && a->nb[1] == x->nb[1] && a->nb[2] == x->nb[2] && a->nb[3] == x->nb[3])) return;
    if (t_overlap(a, wv) || t_overlap(a, bv)) return;
    c.last = a; c.w = (const float *) wv->data; c.b = (const float *) bv->data; c.end = j2;
}

int run_ln_chain(mi_backend_ctx * b, const ln_chain & c, const ggml_cgraph * g) {
    float eps; memcpy(&eps, c.norm->op_params, sizeof(float));
    mi355x_tensor mx = to_mt(c.norm->src[0]), md = to_mt(c.last);
    // encoder / prompt: the next node is an MFMA GEMM on this LayerNorm's result -> write its prepared f16 activations in the same
    // pass (one launch and one read of the 7.7 MB result less per LayerNorm; bit-identical to mi355x_prep_act on the result)
    constexpr bool fuse_prep = true;
    int mode = 0;
    const int j = g ? next_real(g, c.end) : 0;
    if (g && b->fuse && fuse_prep && j < g->n_nodes && mm_takes_prepared(b, g->nodes[j], c.last, mode)) {
        const int64_t K = c.last->ne[0], T = c.last->ne[1];
        const int rr = mi_act_reserve(b, (size_t) T * K * 2);
        if (rr > 0) return rr;
        if (rr == 0) {
            const int rc = mi355x_norm_prep(b->k, &mx, &md, eps, c.w, c.b, b->act, mode);
            if (rc == 0) { b->act_src = c.last->data; b->act_K = K; b->act_T = T; b->act_mode = mode; b->act_nb1 = (int64_t) c.last->nb[1]; return 0; }
            if (rc != MI355X_E_UNSUPPORTED) return rc;
        }
    }
    b->act_src = nullptr;
    return mi355x_norm(b->k, &mx, &md, eps, c.w, c.b);
}


const mi_test_fault & mi_fault() {
    static const mi_test_fault f = [] {
        mi_test_fault t;
        const char * e = getenv("GGML_MI355X_TEST_FAULT");
        if (e && !strncmp(e, "xattn:", 6)) { int l = 0; float x = 1.0f; if (sscanf(e + 6, "%d:%f", &l, &x) == 2) { t.layer = l; t.factor = x; } }
        if (e && !strncmp(e, "reject:", 7)) t.reject_chain = atoi(e + 7);
        return t;
    }();
    return f;
}
// epilogue scale of the output projection that consumes FLASH_ATTN_EXT node i of graph g (1 = untouched)
float mi_fault_scale(const ggml_cgraph * g, int i) {
    const mi_test_fault & f = mi_fault();
    if (f.layer == -2 || g->nodes[i]->src[3]) return 1.0f;
    int ord = 0;
    for (int j = 0; j < i; j++) if (g->nodes[j]->op == GGML_OP_FLASH_ATTN_EXT && !g->nodes[j]->src[3]) ord++;
    return (f.layer < 0 || f.layer == ord) ? f.factor : 1.0f;
}
void mi_fault_apply(const ggml_cgraph * g, int i, mi355x_epilogue & ep) {
    const float fs = mi_fault_scale(g, i);
    if (fs != 1.0f) { ep.scale = (ep.has_scale ? ep.scale : 1.0f) * fs; ep.has_scale = 1; }
}


// is this chain the vocabulary projection whose rows the caller reads back (src/whisper.cpp:2957-2963)?  Then its rows are mirrored.
// (whisper does not flag the logits as a graph output; they are the LAST node of the decoder graph, src/whisper.cpp:2827-2840)
bool mirror_wanted(const ggml_cgraph * g, const mm_chain & ch, int64_t T) {
    const ggml_tensor * l = ch.last;
    return g_mirror_on && ch.end == g->n_nodes - 1 && l == ch.mm && l->type == GGML_TYPE_F32 && l->ne[0] > 8192 && (int64_t) l->nb[1] == l->ne[0]*4 &&
           l->ne[2] == 1 && l->ne[3] == 1 && T >= 1 && T <= MI355X_MAX_COLS && (size_t) (l->ne[0]*4*T) <= MI_MIRROR_CAP - 64;
}

// decoder step: LayerNorm fused into the mat-vec products that consume it (Q/K/V, cross-Q, fc1)
bool try_ln_gemv(mi_backend_ctx * b, const ggml_cgraph * g, const ln_chain & ln, int & end_out, int & rc_out) {
    if (!ln.w || !ln.b) return false;
    const ggml_tensor * x = ln.norm->src[0], * lnout = ln.last;
    const int64_t K = x->ne[0], T = ggml_nrows(x);
    if (T > 8 || K > 2048 || K % 4 || x->ne[2] != 1 || x->ne[3] != 1 || x->nb[0] != 4 || (x->nb[1] % 16) || ((uintptr_t) x->data % 16)) return false;
    if (((uintptr_t) ln.w % 16) || ((uintptr_t) ln.b % 16)) return false;
    const int nuse = use_count(g, lnout);
    if (nuse < 1 || nuse > 3 || lnout->view_src || (lnout->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    mm_chain ch[3];
    int j = next_real(g, ln.end), n = 0;
    while (n < nuse && j < g->n_nodes) {
        const ggml_tensor * t = g->nodes[j];
        if (t->op != GGML_OP_MUL_MAT || t->src[1] != lnout) return false;
        if (!parse_mm_chain(g, j, true, ch[n])) return false;
        j = next_real(g, ch[n].end);
        n++;
    }
    if (n != nuse) return false;
    mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
    d.x = (const float *) x->data; d.x_nb1 = (int64_t) x->nb[1]; d.K = (int) K; d.T = (int) T;
    d.has_norm = 1; memcpy(&d.eps, ln.norm->op_params, sizeof(float)); d.ln_w = ln.w; d.ln_b = ln.b; d.nseg = n;
    for (int s = 0; s < n; s++) {
        const ggml_tensor * w = ch[s].mm->src[0];
        if (w->type != ch[0].mm->src[0]->type || ggml_n_dims(w) > 2 || w->ne[0] != K) return false;
        if (!((is_quant_type(w->type) && ggml_is_contiguous(w)) || (w->type == GGML_TYPE_F16 && ggml_is_contiguous(w)))) return false;
        if (t_overlap(ch[s].last, x)) return false;
        for (int s2 = 0; s2 < s; s2++) if (t_overlap(ch[s].last, ch[s2].last)) return false;
        mi355x_gemv_seg & sg = d.seg[s];
        sg.w = w->data; sg.wtype = (int32_t) w->type; sg.N = (int32_t) w->ne[1]; sg.ep = ch[s].ep;
        sg.dst = ch[s].last->data; sg.dst_type = (int32_t) ch[s].last->type;
        sg.dst_nb1 = ch[s].last->type == GGML_TYPE_F16 ? (int64_t) w->ne[1]*2 : (ch[s].last == ch[s].mm ? (int64_t) ch[s].mm->nb[1] : (int64_t) ch[s].last->nb[1]);
    }
    // (A one-launch "LN + Q projection + cross-attention" kernel existed in round 1.  Re-measured with plain launches it LOSES to the
    //  two launches it replaced, 1.478 -> 1.435 ms/token with it switched off (profiles/r02_decode_env_sweep_final.txt): 64 rows of
    //  W_q per workgroup serialise what 256 workgroups otherwise do in parallel, and a dependent boundary costs only ~1.5 us.  Removed.)
    // (r03 experiment, removed again: LayerNorm + Q/K/V + cache stores + the flash_attn_ext that follows as ONE launch with a workgroup
    //  per head — no hand-off between workgroups, bit-identical records, one dependent launch less per layer.  It LOST: 13.9 us against
    //  4.4 + 4.3 us stand-alone, 351-353 vs 349 ms per chunk (profiles/r03b_head_kbench.txt, r03b_head_check.txt): a head's 192 rows of
    //  integer dots are VALU-bound on ONE CU, ~3 us where 240 workgroups need 0.2.  The kernel is commit 0693a28.)
    mi355x_gemv_cols mcols;
    if (n == 1 && mirror_wanted(g, ch[0], T)) {
        if (char * md = mi_mirror_dev(b)) {
            memset(&mcols, 0, sizeof(mcols));
            for (int t = 0; t < (int) T; t++) mcols.mirror[t] = md + (size_t) t * (size_t) ch[0].last->ne[0] * 4;
            d.cols = &mcols;
        }
    }
    const int rc = mi355x_gemv_fused(b->k, &d);
    if (rc == MI355X_E_UNSUPPORTED) return false;
    if (rc == 0 && d.cols && mi355x_last_launch_mirrored(b->k)) {
        b->mirror_src = ch[0].last->data; b->mirror_bytes = (size_t) ch[0].last->ne[0] * 4 * (size_t) T; b->mirror_state.store(1);
    }
    rc_out = rc; end_out = ch[n - 1].end;
    return true;
}


// decoder step: flash_attn_ext (T <= 8) -> reshape -> mul_mat chain (the O-projection).  The attention kernel leaves
// per-128-key partial records; their combine runs in the prologue of the projection mat-vec (one kernel less per
// attention, src/whisper.cpp:2623-2660 and :2703-2770)
bool try_fattn_gemv(mi_backend_ctx * b, const ggml_cgraph * g, int i, int & end_out, int & rc_out) {
    const ggml_tensor * fa = g->nodes[i];
    const ggml_tensor * q = fa->src[0], * k = fa->src[1], * v = fa->src[2], * m = fa->src[3];
    const int64_t T = q->ne[1];
    if (T > 8 || q->ne[3] != 1 || fa->type != GGML_TYPE_F32 || !ggml_is_contiguous(fa)) return false;
    mi355x_tensor mq = to_mt(q), mk = to_mt(k), mv = to_mt(v), mm_;
    if (m) mm_ = to_mt(m);
    float scale; memcpy(&scale, fa->op_params, 4);
    mi355x_attn_partials parts;
    int rc = mi355x_flash_attn_partial(b->k, &mq, &mk, &mv, m ? &mm_ : nullptr, scale, &parts);
    if (rc == MI355X_E_UNSUPPORTED) return false;
    rc_out = rc; end_out = i;
    if (rc) return true;
    attn_consume(b, g, i, parts, end_out, rc_out);
    return true;
}

// the partial records of flash_attn_ext node i exist: either the projection that follows consumes them (combine in its
// prologue), or they are combined into the node's own memory
void attn_consume(mi_backend_ctx * b, const ggml_cgraph * g, int i, const mi355x_attn_partials & parts, int & end_out, int & rc_out) {
    const ggml_tensor * fa = g->nodes[i];
    const int64_t T = fa->src[0]->ne[1], H = fa->src[0]->ne[2];
    int rc;
    end_out = i;
    bool fused = false;
    const int j = next_real(g, i);
    mm_chain ch;
    if (j < g->n_nodes && g->nodes[j]->op == GGML_OP_MUL_MAT && can_elide(g, fa, 1)) {
        const ggml_tensor * mm = g->nodes[j], * x = mm->src[1], * w = mm->src[0];
        const bool x_is_fa = x->view_src == fa && x->view_offs == 0 && x->type == GGML_TYPE_F32 && ggml_is_contiguous(x) &&
                             x->ne[0] == H*64 && x->ne[1] == T && x->ne[2] == 1 && x->ne[3] == 1 &&
                             use_count(g, x) == 1 && !(x->flags & GGML_TENSOR_FLAG_OUTPUT);
        if (x_is_fa && is_quant_type(w->type) && ggml_is_contiguous(w) && ggml_n_dims(w) <= 2 && w->ne[0] == H*64 &&
            parse_mm_chain(g, j, true, ch)) {
            mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
            d.K = (int) (H*64); d.T = (int) T; d.nseg = 1;
            d.attn_part_o = parts.part_o; d.attn_part_ml = parts.part_ml; d.attn_nparts = parts.nparts;
            mi355x_gemv_seg & sg = d.seg[0];
            sg.w = w->data; sg.wtype = (int32_t) w->type; sg.N = (int32_t) w->ne[1]; sg.ep = ch.ep;
            mi_fault_apply(g, i, sg.ep);
            sg.dst = ch.last->data; sg.dst_type = (int32_t) ch.last->type;
            sg.dst_nb1 = ch.last->type == GGML_TYPE_F16 ? (int64_t) w->ne[1]*2 : (ch.last == ch.mm ? (int64_t) ch.mm->nb[1] : (int64_t) ch.last->nb[1]);
            rc = mi355x_gemv_fused(b->k, &d);
            if (rc != MI355X_E_UNSUPPORTED) { fused = true; rc_out = rc; end_out = ch.end; }
        }
    }
    if (!fused) {
        mi355x_tensor md = to_mt(fa);
        rc_out = mi355x_flash_attn_combine(b->k, &parts, &md);
    }
}

// ---------------------------------------------------------------------------------------------------
// supports_op / single-node dispatch
// ---------------------------------------------------------------------------------------------------
bool whole_quant_ok(const ggml_tensor * t) {   // quantized operands must be whole contiguous tensors (planar layout)
    return !is_quant_type(t->type) || (ggml_is_contiguous(t) && (!t->view_src || (t->view_offs == 0 && ggml_nbytes(t) == ggml_nbytes(t->view_src))));
}

bool mi_supports_op_impl(const ggml_tensor * op) {
    const ggml_tensor * s0 = op->src[0], * s1 = op->src[1];
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            if (op->type != GGML_TYPE_F32) return false;
            const ggml_type wt = s0->type;
            if (!(wt == GGML_TYPE_F32 || wt == GGML_TYPE_F16 || wt == GGML_TYPE_Q4_0 || wt == GGML_TYPE_Q5_0 || wt == GGML_TYPE_Q8_0 || wt == GGML_TYPE_Q4_K)) return false;
            if (s1->type != GGML_TYPE_F32 && s1->type != GGML_TYPE_F16) return false;
            // src1 strided along k (a transposed view: the voice-activity LSTM, src/whisper.cpp:4598-4602): the generic kernel only
            if (s1->nb[0] != ggml_type_size(s1->type)) return !is_quant_type(wt) && s0->nb[0] == ggml_type_size(wt) && s1->nb[0] % ggml_type_size(s1->type) == 0;
            if (is_quant_type(wt)) return whole_quant_ok(s0) && s0->ne[0] % 32 == 0;
            return s0->nb[0] == ggml_type_size(wt);
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            const ggml_tensor * k = s1, * v = op->src[2], * m = op->src[3];
            if (op->src[4]) return false;                                        // sinks
            float max_bias, softcap; memcpy(&max_bias, (const float *) op->op_params + 1, 4); memcpy(&softcap, (const float *) op->op_params + 2, 4);
            if (max_bias != 0.0f || softcap != 0.0f) return false;
            if (s0->type != GGML_TYPE_F32 || k->type != GGML_TYPE_F16 || v->type != GGML_TYPE_F16 || op->type != GGML_TYPE_F32) return false;
            if (s0->ne[0] != 64 || k->ne[0] != 64 || v->ne[0] != 64) return false;
            if (s0->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return false;
            if (s0->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2) return false;
            if ((s0->nb[1] | s0->nb[2] | k->nb[1] | k->nb[2] | v->nb[1] | v->nb[2]) % 16) return false;
            if (m && (m->type != GGML_TYPE_F16 || m->ne[2] != 1 || m->ne[3] != 1 || m->nb[0] != 2)) return false;
            return true;
        }
        case GGML_OP_ADD: case GGML_OP_MUL:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && s1->type == GGML_TYPE_F32 && ggml_are_same_shape(op, s0);
        case GGML_OP_SCALE:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && ggml_is_contiguous(op) && ggml_is_contiguous(s0);
        case GGML_OP_UNARY: {
            // GELU: the whisper graphs; ReLU / sigmoid / tanh: the voice-activity-detection graph (src/whisper.cpp:4545-4680)
            const ggml_unary_op u = ggml_get_unary_op(op);
            return (u == GGML_UNARY_OP_GELU || u == GGML_UNARY_OP_RELU || u == GGML_UNARY_OP_SIGMOID || u == GGML_UNARY_OP_TANH) &&
                   op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && ggml_is_contiguous(op) && ggml_is_contiguous(s0);
        }
        case GGML_OP_SQRT:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && ggml_is_contiguous(op) && ggml_is_contiguous(s0);
        case GGML_OP_PAD_REFLECT_1D:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && op->op_params[0] < s0->ne[0] && op->op_params[1] < s0->ne[0];
        case GGML_OP_NORM:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && s0->nb[0] == 4 && op->nb[0] == 4;
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP:
            return (s0->type == GGML_TYPE_F32 || s0->type == GGML_TYPE_F16) && (op->type == GGML_TYPE_F32 || op->type == GGML_TYPE_F16);
        case GGML_OP_GET_ROWS:
            if (op->type != GGML_TYPE_F32 || s1->type != GGML_TYPE_I32) return false;
            if (is_quant_type(s0->type)) return whole_quant_ok(s0) && s0->ne[2] == 1 && s0->ne[3] == 1 && (op->nb[1] % 16 == 0);
            return s0->type == GGML_TYPE_F32 || s0->type == GGML_TYPE_F16;
        case GGML_OP_IM2COL: {
            const bool is_2d = op->op_params[6] == 1;
            return !is_2d && s1->type == GGML_TYPE_F32 && s1->nb[0] == 4 && (op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_F32) && s1->ne[3] == 1;
        }
        case GGML_OP_SOFT_MAX:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && !op->src[2] && s0->nb[0] == 4 &&
                   (!s1 || ((s1->type == GGML_TYPE_F32 || s1->type == GGML_TYPE_F16) && s1->ne[0] == s0->ne[0]));
        case GGML_OP_ROPE: {
            const int mode = op->op_params[2];
            if (mode == 24 && op->op_params[1] != s0->ne[0] / 2) return false;
            return (mode == 0 || mode == 2 || mode == 8 || mode == 24 || mode == 40) && op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && s0->nb[0] == 4 && op->op_params[15] == 0;
        }
        case GGML_OP_CONCAT:
            return op->type == GGML_TYPE_F32 && s0->type == GGML_TYPE_F32 && s1->type == GGML_TYPE_F32;
        default:
            return false;
    }
}

int run_node(mi_backend_ctx * b, const ggml_tensor * n) {
    mi355x_ctx * k = b->k;
    switch (n->op) {
        case GGML_OP_ADD: case GGML_OP_MUL: {
            mi355x_tensor a = to_mt(n->src[0]), c = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_binary(k, n->op == GGML_OP_ADD ? 0 : 1, &a, &c, &d);
        }
        case GGML_OP_SCALE: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            return mi355x_scale(k, &a, &d, ggml_get_op_params_f32(n, 0), ggml_get_op_params_f32(n, 1));
        }
        case GGML_OP_UNARY: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            switch (ggml_get_unary_op(n)) {
                case GGML_UNARY_OP_GELU:    return mi355x_gelu(k, &a, &d);
                case GGML_UNARY_OP_RELU:    return mi355x_unary(k, MI355X_UNARY_RELU, &a, &d);
                case GGML_UNARY_OP_SIGMOID: return mi355x_unary(k, MI355X_UNARY_SIGMOID, &a, &d);
                case GGML_UNARY_OP_TANH:    return mi355x_unary(k, MI355X_UNARY_TANH, &a, &d);
                default: return MI355X_E_UNSUPPORTED;
            }
        }
        case GGML_OP_SQRT: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            return mi355x_unary(k, MI355X_UNARY_SQRT, &a, &d);
        }
        case GGML_OP_PAD_REFLECT_1D: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            return mi355x_pad_reflect_1d(k, &a, &d, n->op_params[0], n->op_params[1]);
        }
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
            mi355x_tensor a = to_mt(n->src[0]), d = to_mt(n);
            return mi355x_cpy(k, &a, &d);
        }
        case GGML_OP_GET_ROWS: {
            mi355x_tensor a = to_mt(n->src[0]), i = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_get_rows(k, &a, &i, &d);
        }
        case GGML_OP_IM2COL: {
            mi355x_tensor x = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_im2col_1d(k, &x, &d, (int) n->src[0]->ne[0], n->op_params[0], n->op_params[2], n->op_params[4]);
        }
        case GGML_OP_SOFT_MAX: {
            mi355x_tensor x = to_mt(n->src[0]), d = to_mt(n), m;
            if (n->src[1]) m = to_mt(n->src[1]);
            return mi355x_soft_max(k, &x, n->src[1] ? &m : nullptr, &d, ggml_get_op_params_f32(n, 0), ggml_get_op_params_f32(n, 1));
        }
        case GGML_OP_ROPE: {
            mi355x_rope_params p;
            p.n_dims = n->op_params[1]; p.mode = n->op_params[2]; p.n_ctx_orig = n->op_params[4];
            memcpy(&p.freq_base, n->op_params + 5, 4); memcpy(&p.freq_scale, n->op_params + 6, 4); memcpy(&p.ext_factor, n->op_params + 7, 4);
            memcpy(&p.attn_factor, n->op_params + 8, 4); memcpy(&p.beta_fast, n->op_params + 9, 4); memcpy(&p.beta_slow, n->op_params + 10, 4);
            memcpy(p.sections, n->op_params + 11, sizeof(int32_t) * 4);
            mi355x_tensor x = to_mt(n->src[0]), pos = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_rope(k, &x, &pos, n->src[2] ? (const float *) n->src[2]->data : nullptr, &d, &p);
        }
        case GGML_OP_CONCAT: {
            mi355x_tensor a = to_mt(n->src[0]), c = to_mt(n->src[1]), d = to_mt(n);
            return mi355x_concat(k, &a, &c, &d, n->op_params[0]);
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            mi355x_tensor q = to_mt(n->src[0]), kk = to_mt(n->src[1]), v = to_mt(n->src[2]), d = to_mt(n), m;
            if (n->src[3]) m = to_mt(n->src[3]);
            float scale; memcpy(&scale, n->op_params, 4);
            return mi355x_flash_attn_ext(k, &q, &kk, &v, n->src[3] ? &m : nullptr, &d, scale);
        }
        default:
            return MI355X_E_UNSUPPORTED;
    }
}

// ---- the plane pipeline's stages (structures: mi_colset / mi_qstate above) -------------------------------------------
const ggml_tensor * cs_tensor(const mi_colset & cs, int c, int node, int slot) {
    const ggml_tensor * n = (cs.S > 1 ? cs.g[c] : cs.g[0])->nodes[node];
    return slot < 0 ? n : n->src[slot];
}
// address of column c of the tensor at (node, src slot or -1); nb1 < 0: the tensor's own column stride
char * cs_col(const mi_colset & cs, int c, int node, int slot, int64_t nb1) {
    const ggml_tensor * t = cs_tensor(cs, c, node, slot);
    if (cs.S > 1) return (char *) t->data;
    return (char *) t->data + (int64_t) c * (nb1 >= 0 ? nb1 : (int64_t) t->nb[1]);
}
bool q_weight_ok(const ggml_tensor * w, int64_t K) {
    return is_quant_type(w->type) && ggml_is_contiguous(w) && ggml_n_dims(w) <= 2 && w->ne[0] == K && K % 32 == 0 && (w->type != GGML_TYPE_Q4_K || K % 256 == 0);
}
// segment s of a plane mat-vec from mul_mat chain `ch` (graph 0 describes the shapes, every column's graph its own addresses)
void q_fill_seg(const mi_colset & cs, const mm_chain & ch, int s, mi355x_gemv_desc & d, mi355x_gemv_cols & cols) {
    const ggml_tensor * w = ch.mm->src[0];
    mi355x_gemv_seg & sg = d.seg[s];
    sg.w = w->data; sg.wtype = (int32_t) w->type; sg.N = (int32_t) w->ne[1]; sg.ep = ch.ep;
    sg.ep.residual = nullptr; sg.ep.residual_nb1 = 0;                      // per column, below
    sg.dst = nullptr; sg.dst_type = (int32_t) ch.last->type; sg.dst_nb1 = 0;
    const int64_t dst_nb1 = ch.last->type == GGML_TYPE_F16 ? (int64_t) w->ne[1]*2 : (ch.last == ch.mm ? (int64_t) ch.mm->nb[1] : (int64_t) ch.last->nb[1]);
    for (int c = 0; c < cs.T; c++) {
        cols.dst[s][c] = cs_col(cs, c, ch.end, -1, dst_nb1);
        cols.res[s][c] = ch.res_node >= 0 ? (const float *) cs_col(cs, c, ch.res_node, ch.res_slot) : nullptr;
    }
}


// norm -> mul -> add -> 1..3 mul_mat chains on the result (Q/K/V, cross-Q, fc1, logits).  k == nullptr: pattern check only.
bool q_ln_gemv(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, const ln_chain & ln, int & end_out, int & rc_out) {
    const ggml_cgraph * g = cs.g[0];
    if (!ln.w || !ln.b) return false;
    const ggml_tensor * x = ln.norm->src[0], * lnout = ln.last;
    const int64_t K = x->ne[0];
    if (ggml_nrows(x) != (cs.S > 1 ? 1 : cs.T) || K > 2048 || K % 32 || x->ne[2] != 1 || x->ne[3] != 1 || x->nb[0] != 4 || (x->nb[1] % 16) || ((uintptr_t) x->data % 16)) return false;
    if (((uintptr_t) ln.w % 16) || ((uintptr_t) ln.b % 16)) return false;
    const int nuse = use_count(g, lnout);
    if (nuse < 1 || nuse > 3 || lnout->view_src || (lnout->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    mm_chain ch[3];
    int j = next_real(g, ln.end), n = 0;
    while (n < nuse && j < g->n_nodes) {
        const ggml_tensor * t = g->nodes[j];
        if (t->op != GGML_OP_MUL_MAT || t->src[1] != lnout) return false;
        if (!parse_mm_chain(g, j, true, ch[n])) return false;
        j = next_real(g, ch[n].end);
        n++;
    }
    if (n != nuse) return false;
    for (int s = 0; s < n; s++) {
        const ggml_tensor * w = ch[s].mm->src[0];
        if (w->type != ch[0].mm->src[0]->type || !q_weight_ok(w, K)) return false;
        if (t_overlap(ch[s].last, x)) return false;
        for (int s2 = 0; s2 < s; s2++) if (t_overlap(ch[s].last, ch[s2].last)) return false;
    }
    // fc1 + GELU whose only reader is the next mat-vec: the epilogue writes that mat-vec's planes (and skips the F32 store if it can)
    bool pout = false, only = false;
    const ggml_tensor * w0 = ch[0].mm->src[0];
    if (n == 1 && w0->ne[1] % 32 == 0 && w0->ne[1] <= 8192 && w0->type != GGML_TYPE_Q4_K && ch[0].last->type == GGML_TYPE_F32 && ch[0].res_node < 0) {
        const int jn = next_real(g, ch[0].end);
        if (jn < g->n_nodes && g->nodes[jn]->op == GGML_OP_MUL_MAT && g->nodes[jn]->src[1] == ch[0].last) {
            // only if the consumer WILL take the planes (q_mm's own conditions): an elided F32 result exists nowhere else
            const ggml_tensor * w2 = g->nodes[jn]->src[0];
            mi_qstate dummy; int e2 = 0, r2 = 0;
            if (w2->type != GGML_TYPE_Q4_K && q_mm(nullptr, cs, dummy, jn, e2, r2)) { pout = true; only = can_elide(g, ch[0].last, 1); }
        }
    }
    end_out = ch[n - 1].end; rc_out = 0;
    if (!k) return true;
    // k_act_prepare -> planes -> mat-vec: two launches.  The LayerNorm in the mat-vec's own prologue (one launch) lost every time it was built: a
    // workgroup then normalises and quantizes ALL T columns (k_act_prepare spreads them over T workgroups) — k_gemv_q form r03: LN + Q/K/V 15.7 us
    // against 4.5 + 7.2; matrix-core form r05: 16 / 32 streams 10.6 / 7.8 chunks/s against 14.6 / 20.7, beam step 0.552 against 0.487 ms per token
    // (profiles/r05_ln_fused_ab.txt).  Both forms are deleted.
    mi355x_act_desc a; memset(&a, 0, sizeof(a));
    a.K = (int) K; a.T = cs.T; a.wtype = (int32_t) w0->type; a.has_norm = 1; memcpy(&a.eps, ln.norm->op_params, sizeof(float)); a.ln_w = ln.w; a.ln_b = ln.b;
    for (int c = 0; c < cs.T; c++) a.xcol[c] = (const float *) cs_col(cs, c, i, 0);
    void * p0 = mi355x_act_scratch(k, 0), * p1 = mi355x_act_scratch(k, 1);
    if (!p0 || !p1) { rc_out = (int) hipErrorOutOfMemory; return true; }
    int rc = 0;
    mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
    mi355x_gemv_cols cols; memset(&cols, 0, sizeof(cols));
    d.K = (int) K; d.T = cs.T; d.nseg = n; d.cols = &cols;
    for (int s = 0; s < n; s++) q_fill_seg(cs, ch[s], s, d, cols);
    if (pout) { d.planes_out = p1; d.planes_out_only = only ? 1 : 0; }
    auto use_planes = [&]() -> int {
        const int r = mi355x_act_prepare(k, &a, p0);
        if (r) return r;
        d.x_planes = p0;
        return 0;
    };
    rc = use_planes();
    if (rc == MI355X_E_UNSUPPORTED) return false;
    if (rc) { rc_out = rc; return true; }
    bool mirror = n == 1 && cs.owner[0] && mirror_wanted(g, ch[0], cs.S > 1 ? 1 : cs.T);
    if (mirror) {
        const size_t rowb = (size_t) ch[0].last->ne[0] * 4;
        for (int c = 0; c < cs.T && mirror; c++) {
            mi_backend_ctx * ob = cs.S > 1 ? cs.owner[c] : cs.owner[0];
            char * md = ob ? mi_mirror_dev(ob) : nullptr;
            if (!md) mirror = false; else cols.mirror[c] = cs.S > 1 ? md : md + (size_t) c * rowb;
        }
        if (!mirror) memset(cols.mirror, 0, sizeof(cols.mirror));
    }
    rc = mi355x_gemv_fused(k, &d);
    if (rc == 0 && mirror && mi355x_last_launch_mirrored(k)) {
        const size_t rowb = (size_t) ch[0].last->ne[0] * 4;
        if (cs.S > 1) for (int c = 0; c < cs.T; c++) { mi_backend_ctx * ob = cs.owner[c]; ob->mirror_src = cs_tensor(cs, c, ch[0].end, -1)->data; ob->mirror_bytes = rowb; ob->mirror_state.store(1); }
        else { mi_backend_ctx * ob = cs.owner[0]; ob->mirror_src = ch[0].last->data; ob->mirror_bytes = rowb * (size_t) cs.T; ob->mirror_state.store(1); }
    }
    if (rc == MI355X_E_UNSUPPORTED && pout) { d.planes_out = nullptr; d.planes_out_only = 0; pout = false; rc = mi355x_gemv_fused(k, &d); }
    if (rc == MI355X_E_UNSUPPORTED) { rc_out = (int) hipErrorInvalidValue; GGML_LOG_ERROR("ggml-mi355x: plane mat-vec rejected a shape its planes were already prepared for\n"); return true; }
    rc_out = rc;
    qs = mi_qstate();
    if (pout && rc == 0) { qs.src = ch[0].last->data; qs.K = w0->ne[1]; qs.T = cs.T; qs.which = 1; }
    return true;
}

// flash_attn_ext -> reshape -> mul_mat chain (output projection)
bool q_attn_proj(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, int & end_out, int & rc_out) {
    const ggml_cgraph * g = cs.g[0];
    const ggml_tensor * fa = g->nodes[i];
    const ggml_tensor * q = fa->src[0], * kk = fa->src[1], * v = fa->src[2], * m = fa->src[3];
    const int64_t Tq = q->ne[1], H = q->ne[2];
    if (Tq != (cs.S > 1 ? 1 : cs.T) || q->ne[3] != 1 || fa->type != GGML_TYPE_F32 || !ggml_is_contiguous(fa) || H*64 > 2048) return false;
    const int j = next_real(g, i);
    if (!(j < g->n_nodes && g->nodes[j]->op == GGML_OP_MUL_MAT && can_elide(g, fa, 1))) return false;
    const ggml_tensor * mm = g->nodes[j], * x = mm->src[1], * w = mm->src[0];
    const bool x_is_fa = x->view_src == fa && x->view_offs == 0 && x->type == GGML_TYPE_F32 && ggml_is_contiguous(x) &&
                         x->ne[0] == H*64 && x->ne[1] == Tq && x->ne[2] == 1 && x->ne[3] == 1 && use_count(g, x) == 1 && !(x->flags & GGML_TENSOR_FLAG_OUTPUT);
    mm_chain ch;
    if (!x_is_fa || !q_weight_ok(w, H*64) || !parse_mm_chain(g, j, true, ch)) return false;
    end_out = ch.end; rc_out = 0;
    if (!k) return true;
    float scale; memcpy(&scale, fa->op_params, 4);
    mi355x_attn_partials parts;
    mi355x_tensor mq = to_mt(q), mk = to_mt(kk), mv = to_mt(v), mm_;
    int rc;
    void * p0 = mi355x_act_scratch(k, 0);
    if (!p0) { rc_out = (int) hipErrorOutOfMemory; return true; }
    // every column's operands (own graph in a cross-state batch; the token columns of the one graph otherwise)
    mi355x_attn_state st[MI355X_MAX_COLS]; memset(st, 0, sizeof(st));
    int max_kv = 0;
    for (int c = 0; c < cs.T; c++) {
        if (cs.S > 1) {
            const ggml_tensor * qc = cs_tensor(cs, c, i, 0), * kc = cs_tensor(cs, c, i, 1), * vc = cs_tensor(cs, c, i, 2), * mc = cs_tensor(cs, c, i, 3);
            if (kc->nb[1] != kk->nb[1] || kc->nb[2] != kk->nb[2] || vc->nb[1] != v->nb[1] || vc->nb[2] != v->nb[2] || qc->nb[2] != q->nb[2] || (mc != nullptr) != (m != nullptr) || vc->ne[1] != kc->ne[1]) {
                rc_out = (int) hipErrorInvalidValue; GGML_LOG_ERROR("ggml-mi355x: cross-state batch: attention operands of the states are laid out differently\n"); return true;
            }
            st[c].q = qc->data; st[c].k = kc->data; st[c].v = vc->data; st[c].mask = mc ? mc->data : nullptr; st[c].n_kv = (int32_t) kc->ne[1];
        } else {
            st[c].q = (const char *) q->data + (int64_t) c * (int64_t) q->nb[1]; st[c].k = kk->data; st[c].v = v->data;
            st[c].mask = m ? (const char *) m->data + (int64_t) c * (int64_t) m->nb[1] : nullptr; st[c].n_kv = (int32_t) kk->ne[1];
        }
        max_kv = std::max(max_kv, (int) st[c].n_kv);
    }
    // ONE launch from q / K / V to the projection's activation planes for self-attention (<= 512 keys).  Cross-attention's 1500 keys the same
    // way (three rounds in one 16-wave workgroup per (head, column)) lost twice — r03: 13.5 us against 6.5 + 6.7 for partial records + combine;
    // r05 with the matrix-core mat-vecs: 16 / 32 streams 14.8 / 18.5 chunks/s against 15.7 / 21.9 (profiles/r05_stream_scaling.txt)
    constexpr int planes_max_kv = 512;
    bool have_planes = false;
    if (max_kv <= planes_max_kv && w->type != GGML_TYPE_Q4_K && (!m || (m->type == GGML_TYPE_F16 && m->nb[0] == 2))) {
        rc = mi355x_flash_attn_planes(k, cs.T, st, &mq, &mk, &mv, scale, p0);
        if (rc == 0) have_planes = true;
        else if (rc != MI355X_E_UNSUPPORTED) { rc_out = rc; return true; }
    }
    if (!have_planes) {
        if (cs.S > 1) rc = mi355x_flash_attn_partial_multi(k, cs.T, st, &mq, &mk, &mv, scale, &parts);
        else {
            if (m) mm_ = to_mt(m);
            rc = mi355x_flash_attn_partial(k, &mq, &mk, &mv, m ? &mm_ : nullptr, scale, &parts);
        }
        if (rc == MI355X_E_UNSUPPORTED) return false;
        if (rc) { rc_out = rc; return true; }
        mi355x_act_desc a; memset(&a, 0, sizeof(a));
        a.K = (int) (H*64); a.T = cs.T; a.wtype = (int32_t) w->type;
        a.attn_part_o = parts.part_o; a.attn_part_ml = parts.part_ml; a.attn_nparts = parts.nparts;
        rc = mi355x_act_prepare(k, &a, p0);
    }
    mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
    mi355x_gemv_cols cols; memset(&cols, 0, sizeof(cols));
    d.K = (int) (H*64); d.T = cs.T; d.nseg = 1; d.x_planes = p0; d.cols = &cols;
    q_fill_seg(cs, ch, 0, d, cols);
    mi_fault_apply(g, i, d.seg[0].ep);
    if (rc == 0) rc = mi355x_gemv_fused(k, &d);
    if (rc == MI355X_E_UNSUPPORTED) { rc = (int) hipErrorInvalidValue; GGML_LOG_ERROR("ggml-mi355x: plane pipeline rejected the attention output projection\n"); }
    rc_out = rc;
    qs = mi_qstate();
    return true;
}

// mul_mat chain on an F32 activation (fc2; any projection the patterns above did not take)
bool q_mm(mi355x_ctx * k, const mi_colset & cs, mi_qstate & qs, int i, int & end_out, int & rc_out) {
    const ggml_cgraph * g = cs.g[0];
    mm_chain ch;
    if (!parse_mm_chain(g, i, true, ch)) return false;
    const ggml_tensor * w = ch.mm->src[0], * x = ch.mm->src[1];
    const int64_t K = w->ne[0];
    if (!q_weight_ok(w, K) || K > 8192 || x->type != GGML_TYPE_F32 || x->ne[1] != (cs.S > 1 ? 1 : cs.T) || x->ne[2] != 1 || x->ne[3] != 1 || x->nb[0] != 4 ||
        (x->nb[1] % 16) || ((uintptr_t) x->data % 16) || ch.mm->type != GGML_TYPE_F32) return false;
    end_out = ch.end; rc_out = 0;
    if (!k) return true;
    void * planes;
    int rc = 0;
    if (qs.src == x->data && qs.K == K && qs.T == cs.T) planes = mi355x_act_scratch(k, qs.which);
    else {
        planes = mi355x_act_scratch(k, 0);
        mi355x_act_desc a; memset(&a, 0, sizeof(a));
        a.K = (int) K; a.T = cs.T; a.wtype = (int32_t) w->type;
        for (int c = 0; c < cs.T; c++) a.xcol[c] = (const float *) cs_col(cs, c, i, 1);
        rc = planes ? mi355x_act_prepare(k, &a, planes) : (int) hipErrorOutOfMemory;
        if (rc == MI355X_E_UNSUPPORTED) return false;
    }
    qs = mi_qstate();
    if (rc) { rc_out = rc; return true; }
    mi355x_gemv_desc d; memset(&d, 0, sizeof(d));
    mi355x_gemv_cols cols; memset(&cols, 0, sizeof(cols));
    d.K = (int) K; d.T = cs.T; d.nseg = 1; d.x_planes = planes; d.cols = &cols;
    q_fill_seg(cs, ch, 0, d, cols);
    rc = mi355x_gemv_fused(k, &d);
    if (rc == MI355X_E_UNSUPPORTED) { rc = (int) hipErrorInvalidValue; GGML_LOG_ERROR("ggml-mi355x: plane mat-vec rejected a shape its planes were already prepared for\n"); }
    rc_out = rc;
    return true;
}

// One launch chain for S single-token decoder graphs (the columns).  k == nullptr: does every node fit?  (nothing is launched)
int mi_walk_batch(mi355x_ctx * k, const mi_colset & cs) {
    const ggml_cgraph * g = cs.g[0];
    mi_qstate qs;
    static std::atomic<int> n_real_walks{0};
    const bool inject_reject = k && mi_fault().reject_chain > 0 && ++n_real_walks == mi_fault().reject_chain;      // (TEST fault injection, see mi_test_fault)
    for (int i = 0; i < g->n_nodes; i++) {
        const ggml_tensor * n = g->nodes[i];
        if (op_is_empty(n) || ggml_is_empty(n) || !(n->flags & GGML_TENSOR_FLAG_COMPUTE)) continue;
        if (inject_reject && i > g->n_nodes / 2) { mi355x_flush(k); GGML_LOG_WARN("ggml-mi355x: TEST fault: merged chain rejected at node %d of %d\n", i, g->n_nodes); return (int) hipErrorInvalidValue; }
        int end = i, rc = 0;
        bool took = false;
        if (n->op == GGML_OP_GET_ROWS) {
            // token embedding + positional embedding of every state in one launch
            const int j1 = next_real(g, i), j2 = j1 < g->n_nodes ? next_real(g, j1) : g->n_nodes;
            if (j2 < g->n_nodes && g->nodes[j1]->op == GGML_OP_GET_ROWS && g->nodes[j2]->op == GGML_OP_ADD) {
                const ggml_tensor * ga = n, * gb = g->nodes[j1], * ad = g->nodes[j2];
                const bool pair = (ad->src[0] == ga && ad->src[1] == gb) || (ad->src[0] == gb && ad->src[1] == ga);
                if (pair && can_elide(g, ga, 1) && can_elide(g, gb, 1) && ggml_are_same_shape(ga, gb) && ggml_are_same_shape(ad, ga) && ad->type == GGML_TYPE_F32 &&
                    ggml_is_contiguous(ad) && gb->src[0]->type == GGML_TYPE_F32 && ga->src[1]->type == GGML_TYPE_I32 && gb->src[1]->type == GGML_TYPE_I32 &&
                    ggml_nelements(ga->src[1]) == 1 && ggml_nelements(gb->src[1]) == 1 && whole_quant_ok(ga->src[0])) {
                    took = true; end = j2;
                    if (k) {
                        mi355x_head_state st[MI355X_MAX_COLS]; memset(st, 0, sizeof(st));
                        for (int c = 0; c < cs.T; c++) {
                            st[c].tok = (const int32_t *) cs_tensor(cs, c, i, 1)->data; st[c].pos = (const int32_t *) cs_tensor(cs, c, j1, 1)->data;
                            st[c].dst = (float *) cs_tensor(cs, c, j2, -1)->data;
                        }
                        mi355x_tensor te = to_mt(ga->src[0]), pe = to_mt(gb->src[0]);
                        rc = mi355x_decode_head_multi(k, cs.T, st, &te, &pe);
                    }
                }
            }
        } else if (n->op == GGML_OP_CPY || n->op == GGML_OP_CONT || n->op == GGML_OP_DUP) {
            // the mask row's F32 -> F16 cast (src/whisper.cpp:2520), every state's in one launch
            const ggml_tensor * s0 = n->src[0];
            if (s0->type == GGML_TYPE_F32 && n->type == GGML_TYPE_F16 && ggml_is_contiguous(s0) && ggml_is_contiguous(n) && ggml_nelements(s0) == ggml_nelements(n) &&
                ggml_nelements(n) < (1 << 20)) {
                took = true;
                if (k) {
                    mi355x_head_state st[MI355X_MAX_COLS]; memset(st, 0, sizeof(st));
                    for (int c = 0; c < cs.T; c++) {
                        const ggml_tensor * nc = cs_tensor(cs, c, i, -1);
                        st[c].mask_f32 = (const float *) nc->src[0]->data; st[c].mask_f16 = nc->data; st[c].n_mask = (int32_t) ggml_nelements(nc);
                    }
                    rc = mi355x_decode_head_multi(k, cs.T, st, nullptr, nullptr);          // no state embeds: the tables are not needed
                }
            }
        } else if (n->op == GGML_OP_NORM) {
            ln_chain c;
            parse_ln_chain(g, i, true, c);
            took = q_ln_gemv(k, cs, qs, i, c, end, rc);
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT) {
            took = q_attn_proj(k, cs, qs, i, end, rc);
        } else if (n->op == GGML_OP_MUL_MAT) {
            took = q_mm(k, cs, qs, i, end, rc);
        }
        if (!took) return MI355X_E_UNSUPPORTED;
        if (rc != 0) { GGML_LOG_ERROR("ggml-mi355x: cross-state batch: op %s (%s) failed: rc=%d %s\n", ggml_op_name(n->op), n->name, rc, mi355x_last_error()); return rc; }
        i = end;
    }
    return k ? mi355x_flush(k) : 0;
}

// may graph `b` run as another column next to graph `a`?  Same node sequence, shapes (up to the key counts) and weights.
bool mi_graphs_congruent(const ggml_cgraph * a, const ggml_cgraph * b) {
    if (a->n_nodes != b->n_nodes) return false;
    for (int i = 0; i < a->n_nodes; i++) {
        const ggml_tensor * x = a->nodes[i], * y = b->nodes[i];
        // (extents are not compared: the key count n_kv — mask rows, K / V views — legitimately differs between states that are at
        //  different positions; with equal ops, types and WEIGHTS every other extent follows from the model)
        if (x->op != y->op || x->type != y->type || (x->flags & GGML_TENSOR_FLAG_COMPUTE) != (y->flags & GGML_TENSOR_FLAG_COMPUTE)) return false;
        for (int s = 0; s < 4; s++) {
            const ggml_tensor * xs = x->src[s], * ys = y->src[s];
            if ((xs == nullptr) != (ys == nullptr)) return false;
            if (!xs) continue;
            if (xs->type != ys->type) return false;
            ggml_backend_buffer_t xb = xs->view_src ? xs->view_src->buffer : xs->buffer;
            if (xb && xb->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS && xs->data != ys->data) return false;     // the same weights
        }
    }
    return true;
}

// walk nodes [i0, i_stop) and emit kernels on the backend's stream
int mi_emit_range(mi_backend_ctx * b, ggml_cgraph * g, int i0, int i_stop) {
    int i = i0;
    const uint64_t trace_n0 = g_trace ? mi355x_eager_count(b->k) : 0;
    bool trace_first = g_trace && b->trace_gc_enter != 0;
    for (; i < i_stop; i++) {
        if (trace_first && mi355x_eager_count(b->k) != trace_n0) { g_trace_ns[2] += trace_now() - b->trace_gc_enter; g_trace_calls[2]++; trace_first = false; }
        const ggml_tensor * n = g->nodes[i];
        if (op_is_empty(n) || ggml_is_empty(n) || !(n->flags & GGML_TENSOR_FLAG_COMPUTE)) continue;
        int rc = MI355X_E_UNSUPPORTED;
        // decoder steps with planes_min_t .. 8 columns (beam search): the pre-quantized-activation pipeline (stages above); whatever it
        // does not take falls through to the fused / generic paths below
        constexpr int planes_min_t = 3;
        if (b->fuse && !b->exact && (n->op == GGML_OP_MUL_MAT || n->op == GGML_OP_NORM || n->op == GGML_OP_FLASH_ATTN_EXT)) {
            const int64_t Tn = n->op == GGML_OP_FLASH_ATTN_EXT ? n->src[0]->ne[1] : (n->op == GGML_OP_MUL_MAT ? n->src[1]->ne[1] : ggml_nrows(n->src[0]));
            if (Tn >= planes_min_t && Tn <= MI355X_IMG_COLS) {
                mi_colset cs; cs.S = 1; cs.T = (int) Tn; cs.g[0] = g; cs.owner[0] = b;
                int end = i, rc2 = 0; bool took = false;
                if (n->op == GGML_OP_NORM) { ln_chain c; parse_ln_chain(g, i, true, c); took = q_ln_gemv(b->k, cs, b->qs, i, c, end, rc2); }
                else if (n->op == GGML_OP_FLASH_ATTN_EXT) took = q_attn_proj(b->k, cs, b->qs, i, end, rc2);
                else took = q_mm(b->k, cs, b->qs, i, end, rc2);
                if (took) {
                    if (rc2 != 0) { GGML_LOG_ERROR("ggml-mi355x: op %s (%s) failed in the plane pipeline: rc=%d %s\n", ggml_op_name(n->op), n->name, rc2, mi355x_last_error()); return rc2; }
                    b->act_src = nullptr;
                    i = end;
                    continue;
                }
            }
        }
        b->qs = mi_qstate();
        if (n->op == GGML_OP_MUL_MAT) {
            mm_chain c;
            parse_mm_chain(g, i, b->fuse, c);
            rc = run_mm_chain(b, c, g);
            if (rc == MI355X_E_UNSUPPORTED && c.end != i) { parse_mm_chain(g, i, false, c); rc = run_mm_chain(b, c); }
            else i = c.end;
        } else if (n->op == GGML_OP_NORM) {
            ln_chain c;
            parse_ln_chain(g, i, b->fuse, c);
            int end = 0, rc2 = 0;
            if (b->fuse && try_ln_gemv(b, g, c, end, rc2)) { rc = rc2; i = end; }
            else {
                rc = run_ln_chain(b, c, g);                  // leaves b->act_* describing the prepared activations, if it made them
                if (rc == MI355X_E_UNSUPPORTED && c.end != i) { parse_ln_chain(g, i, false, c); rc = run_ln_chain(b, c); }
                else i = c.end;
            }
        } else if (n->op == GGML_OP_GET_ROWS && b->fuse) {
            // token embedding + positional embedding: get_rows, get_rows, add -> one launch
            rc = MI355X_E_UNSUPPORTED;
            const int j1 = next_real(g, i), j2 = j1 < g->n_nodes ? next_real(g, j1) : g->n_nodes;
            if (j2 < g->n_nodes && g->nodes[j1]->op == GGML_OP_GET_ROWS && g->nodes[j2]->op == GGML_OP_ADD) {
                const ggml_tensor * ga = n, * gb = g->nodes[j1], * ad = g->nodes[j2];
                const bool pair = (ad->src[0] == ga && ad->src[1] == gb) || (ad->src[0] == gb && ad->src[1] == ga);
                if (pair && can_elide(g, ga, 1) && can_elide(g, gb, 1) && ggml_are_same_shape(ga, gb) && ggml_are_same_shape(ad, ga) &&
                    ad->type == GGML_TYPE_F32 && ggml_is_contiguous(ad) && gb->src[0]->type == GGML_TYPE_F32) {
                    mi355x_tensor sa = to_mt(ga->src[0]), ia = to_mt(ga->src[1]), sb = to_mt(gb->src[0]), ib = to_mt(gb->src[1]), d = to_mt(ad);
                    rc = mi355x_get_rows_add(b->k, &sa, &ia, &sb, &ib, &d);
                    if (rc != MI355X_E_UNSUPPORTED) i = j2;
                }
            }
            if (rc == MI355X_E_UNSUPPORTED) rc = run_node(b, n);
            b->act_src = nullptr;
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT && b->exact) {
            mi355x_tensor q = to_mt(n->src[0]), kk = to_mt(n->src[1]), v = to_mt(n->src[2]), d = to_mt(n), m;
            if (n->src[3]) m = to_mt(n->src[3]);
            float scale; memcpy(&scale, n->op_params, 4);
            rc = mi355x_flash_attn_ext_exact(b->k, &q, &kk, &v, n->src[3] ? &m : nullptr, &d, scale, b->n_threads);
            if (rc == MI355X_E_UNSUPPORTED) rc = run_node(b, n);
            b->act_src = nullptr;
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT && b->fuse && n->src[0]->ne[1] > 8) {
            // encoder / prompt attention whose result (through a reshape) is the activation matrix of the output projection: the
            // attention kernel leaves that GEMM's prepared f16 activations as well (one launch and one pass over the result less)
            constexpr bool on = true;
            const int j = next_real(g, i);
            const int64_t T = n->src[0]->ne[1], NS = n->ne[0] * n->ne[1];
            int mode = -1;
            rc = MI355X_E_UNSUPPORTED;
            if (on && j < g->n_nodes && g->nodes[j]->op == GGML_OP_MUL_MAT) {
                const ggml_tensor * x = g->nodes[j]->src[1];
                if (x->data == n->data && x->ne[0] == NS && x->ne[1] == T && (int64_t) x->nb[1] == NS*4 && ggml_is_contiguous(n) &&
                    mm_takes_prepared(b, g->nodes[j], x, mode) && (mode == 1 || mode == 3) && mi_act_reserve(b, (size_t) T * NS * 2) == 0) {
                    mi355x_tensor q = to_mt(n->src[0]), kk = to_mt(n->src[1]), v = to_mt(n->src[2]), d = to_mt(n), m;
                    if (n->src[3]) m = to_mt(n->src[3]);
                    float scale; memcpy(&scale, n->op_params, 4);
                    rc = mode == 3 ? mi355x_flash_attn_ext_prep_rows(b->k, &q, &kk, &v, n->src[3] ? &m : nullptr, &d, scale, b->act)
                                   : mi355x_flash_attn_ext_prep(b->k, &q, &kk, &v, n->src[3] ? &m : nullptr, &d, scale, b->act);
                    if (rc == 0) { b->act_src = x->data; b->act_K = NS; b->act_T = T; b->act_mode = mode; b->act_nb1 = NS*4; }
                }
            }
            if (rc == MI355X_E_UNSUPPORTED) { rc = run_node(b, n); b->act_src = nullptr; }
        } else if (n->op == GGML_OP_FLASH_ATTN_EXT && b->fuse && n->src[0]->ne[1] <= 8) {
            int end = i, rc2 = MI355X_E_UNSUPPORTED;
            if (try_fattn_gemv(b, g, i, end, rc2)) { rc = rc2; i = end; }
            else rc = run_node(b, n);
            b->act_src = nullptr;
        } else {
            rc = run_node(b, n);
            b->act_src = nullptr;
        }
        if (rc != 0) {
            GGML_LOG_ERROR("ggml-mi355x: op %s (%s) failed: rc=%d %s\n", ggml_op_name(n->op), n->name, rc, mi355x_last_error());
            return rc;
        }
    }
    return mi355x_flush(b->k);            // launches the kernel library held back for grouping (gemm_mfma.hip) leave with their range
}
int mi_emit_graph(mi_backend_ctx * b, ggml_cgraph * g) {
    b->act_src = nullptr; b->elided_src = nullptr; b->elided_for = nullptr; b->qs = mi_qstate();
    return mi_emit_range(b, g, 0, g->n_nodes);
}

