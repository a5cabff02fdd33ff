// mi_batching.cpp — part of the MI355X ggml backend plugin; see mi_backend.h for the map of the translation units.
#include "mi_backend.h"


mi_batch_group     g_batch[MI_MAX_DEVICES];
std::atomic<int>   g_batching{-1};            // -1: not decided yet (environment), 0 off, 1 on (from mi_batch_min_states() states), n >= 2: on from n states
bool mi_batching_on() {
    int v = g_batching.load();
    // on by default (r04): fewer than mi_batch_min_states() decoding states keep their own chains anyway, and beyond four states own
    // chains collapse (8 states: 1.8 chunks/s against 9.4 merged) — a whisper_full_parallel user must not have to know a switch
    if (v < 0) { const char * e = getenv("GGML_MI355X_BATCH"); v = e ? std::max(0, atoi(e)) : 1; g_batching.store(v); }
    return v != 0;
}
// Fewer decoding states than this run their own chains side by side (states-on-streams) although batching is on: a merged chain costs
// 12 launches per layer against 8 and moves the states in lockstep (their host phases no longer hide behind each other's GPU work) —
// measured large-v3 Q5_0: 2 / 4 states 3.35 / 5.96 chunks/s merged against 4.1 / 7.5 on their own streams, 8 states 9.75 against 2.8
// (profiles/r03_stream_scaling_*).  ggml_backend_mi355x_set_batching(n >= 2) / GGML_MI355X_BATCH=n sets the threshold to n.
int mi_batch_min_states() {
    constexpr int env_min = 5;
    const int v = g_batching.load();
    return v >= 2 ? v : env_min;
}

// a single-token decoder step?  (cheap signature; whether every node fits is decided once per graph shape by the dry walk)
bool mi_is_step_graph(const ggml_cgraph * g) {
    if (g->n_nodes < 32) return false;
    const ggml_tensor * last = g->nodes[g->n_nodes - 1];
    if (last->op != GGML_OP_MUL_MAT || last->ne[1] != 1 || last->ne[2] != 1 || last->ne[3] != 1) return false;
    for (int i = 0; i < g->n_nodes && i < 16; i++) {
        const ggml_tensor * n = g->nodes[i];
        if (op_is_empty(n)) continue;
        return n->op == GGML_OP_GET_ROWS && ggml_nelements(n->src[1]) == 1;
    }
    return false;
}

void mi_batch_leave(mi_backend_ctx * b) {
    if (!b->in_group) return;
    mi_batch_group & grp = g_batch[b->device];
    std::lock_guard<std::mutex> lk(grp.m);
    if (!b->in_group) return;
    b->in_group = false;
    for (size_t i = 0; i < grp.members.size(); i++) if (grp.members[i] == b) { grp.members.erase(grp.members.begin() + i); break; }
    grp.cv.notify_all();
}


// the merged launch chain for `n` members (group lock NOT held).  Falls back to every member alone when the graphs do not fit.
// returns true when the members left as ONE merged chain
bool mi_compute_batch(mi_batch_group & grp, mi_batch_group::lane & ln, mi_batch_member ** mem, int n) {
    mi_backend_ctx * b0 = mem[0]->b;
    (void) hipSetDevice(b0->device);
    bool ok = true;
    if (!ln.k) {
        ln.k = mi355x_ctx_create(b0->device);
        for (auto & e : ln.ev_ring) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) ok = false;
        if (!ln.k) ok = false;
    }
    mi_colset cs; cs.S = n; cs.T = n;
    for (int c = 0; c < n; c++) { cs.g[c] = mem[c]->g; cs.owner[c] = mem[c]->b; }
    if (ok) {
        const ggml_cgraph * g0 = cs.g[0];
        const uint64_t sn = (uint64_t) g0->n_nodes; const void * sw = g0->nodes[g0->n_nodes - 1]->src[0]->data;
        bool same = true;
        for (int c = 1; c < n; c++) same = same && cs.g[c]->n_nodes == g0->n_nodes && cs.g[c]->nodes[g0->n_nodes - 1]->src[0]->data == sw;
        std::lock_guard<std::mutex> sl(grp.sig_m);
        if (!same) ok = false;
        else {
            // every state is checked ONCE per graph shape — node for node against the chain's first graph — before it may be a column,
            // not only the states that happened to be in the first batch of that shape
            bool fresh = grp.sig_nodes != sn || grp.sig_w != sw;
            // a state already verified for this shape (an unverified state in column 0 used to be compared with itself only)
            int vref = -1;
            for (int c = 0; c < n && vref < 0; c++) if (mem[c]->b->sig_nodes == sn && mem[c]->b->sig_w == sw) vref = c;
            if (vref < 0) fresh = true;                                      // nobody here has been checked: the dry walk below vouches for column 0
            for (int c = 0; c < n && ok; c++) {
                mi_backend_ctx * bc = mem[c]->b;
                if (!fresh && bc->sig_nodes == sn && bc->sig_w == sw) continue;          // verified earlier, and so is the graph it is compared with (transitively)
                const int ref = fresh ? 0 : vref;                           // fresh: everybody against column 0 (walked below); else against a verified member
                if (c != ref) ok = mi_graphs_congruent(cs.g[ref], cs.g[c]);
            }
            if (ok && fresh) ok = mi_walk_batch(nullptr, cs) == 0;
            if (ok) { grp.sig_nodes = sn; grp.sig_w = sw; for (int c = 0; c < n; c++) { mem[c]->b->sig_nodes = sn; mem[c]->b->sig_w = sw; } }
            else for (int c = 0; c < n; c++) mem[c]->b->no_batch_nodes = g0->n_nodes;       // this graph shape never batches: stop joining with it
        }
    }
    if (!ok) {
        grp.n_fallback++;
        for (int c = 0; c < n; c++) mem[c]->status = mi_compute_own(mem[c]->b, mem[c]->g);
        return false;
    }
    hipStream_t bs = (hipStream_t) mi355x_ctx_stream(ln.k);
    for (int c = 0; c < n; c++) {
        mi_backend_ctx * b = mem[c]->b;
        if (b->own_dirty) {                                   // the member's earlier work on its own stream (encoder -> cross-KV, a solo step's KV writes)
            (void) mi355x_flush(b->k);
            if (!b->own_ev) (void) hipEventCreateWithFlags(&b->own_ev, hipEventDisableTiming);
            (void) hipEventRecord(b->own_ev, (hipStream_t) mi355x_ctx_stream(b->k));
            (void) hipStreamWaitEvent(bs, b->own_ev, 0);
            b->own_dirty = false;
        }
        // the previous chain this state was a column of may have run on ANOTHER lane: its KV-cache and activation writes must be
        // ordered in front of this chain by the streams themselves, not by the host synchronize whisper happens to call between steps
        // (graph_compute is an asynchronous entry point)
        if (b->batch_wait_stream) (void) hipStreamWaitEvent(bs, b->batch_wait_stream, 0);
    }
    // the members' graph inputs (token id, position, mask: a few hundred bytes per state), kept until the chain is out: ggml's graph allocator may hand an
    // input's memory to a later node of the same graph, so a step that is repeated after a partial chain must find its inputs staged again
    mi_io_saved inputs;
    {
        const char * base[MI355X_MAX_COLS]; size_t size[MI355X_MAX_COLS]; int nr = 0;
        for (int c = 0; c < n; c++) {
            const ggml_tensor * t0 = cs.g[c]->nodes[0];
            ggml_backend_buffer_t buf = t0->view_src ? t0->view_src->buffer : t0->buffer;
            if (buf) { base[nr] = (const char *) ggml_backend_buffer_get_base(buf); size[nr] = ggml_backend_buffer_get_size(buf); nr++; }
        }
        mi_io_snapshot(b0->device, base, size, nr, inputs);
    }
    mi_io_order_stream(b0->device, ln.io, bs);                // every member's graph inputs leave with one scatter launch at the head of the chain
    const int rc = mi_walk_batch(ln.k, cs);
    if (rc != 0) {
        (void) mi355x_ctx_synchronize(ln.k);
        grp.n_fallback++;
        if (rc != MI355X_E_UNSUPPORTED && rc != (int) hipErrorInvalidValue) {
            // a device fault, not a rejection: repeating the step n times on the states' own chains would hit the same fault n times
            GGML_LOG_ERROR("ggml-mi355x: cross-state batch failed mid-chain (rc=%d %s): %d states report failure\n", rc, mi355x_last_error(), n);
            for (int c = 0; c < n; c++) mem[c]->status = GGML_STATUS_FAILED;
            return false;
        }
        // a kernel rejected the chain half-way (a shape or alignment only its launch code knows): what was launched has written nothing a
        // repeat would not write again (activations, this position's KV rows) — except, possibly, over the memory of graph inputs the allocator
        // reuses for later nodes: those are staged again from the copy taken above.  Every member then runs its step again on its own chain
        // once the partial chain has drained, and this graph shape stops batching for these states
        GGML_LOG_WARN("ggml-mi355x: cross-state batch rejected mid-chain (rc=%d %s): %d states repeat the step on their own chains\n", rc, mi355x_last_error(), n);
        { std::lock_guard<std::mutex> sl(grp.sig_m); grp.sig_nodes = 0; grp.sig_w = nullptr; }
        mi_io_replay(b0->device, inputs);                     // (pending again: the first member's own chain scatters them at its head, the others' streams wait for that flush)
        for (int c = 0; c < n; c++) {
            mem[c]->b->no_batch_nodes = mem[c]->g->n_nodes; mem[c]->b->sig_nodes = 0;
            mem[c]->status = mi_compute_own(mem[c]->b, mem[c]->g);
        }
        return false;
    }
    hipEvent_t ev = ln.ev_ring[ln.ev_next]; ln.ev_next = (ln.ev_next + 1) % 16;
    (void) hipEventRecord(ev, bs);
    for (int c = 0; c < n; c++) {
        mem[c]->b->batch_wait_sync = ev; mem[c]->b->batch_wait_stream = ev;
        mem[c]->status = GGML_STATUS_SUCCESS;
    }
    return true;
}

ggml_status mi_batch_join(mi_backend_ctx * b, ggml_cgraph * cgraph) {
    // (the window must stay above a chain step: 1 ms / 0.4 ms collapse to 2.4 / 2.0 chunks/s at 16 streams — states are dropped while they are simply on
    //  their way through the host part of a step; profiles/r05_stream_scaling.txt)
    constexpr double window_ms = 3.0;
    // columns per merged chain.  GGML_MI355X_BATCH_COLS=n (2..32) fixes it; by default 60 % of the decoding states (at least 4) ride one chain and
    // the rest a second one next to it (MI_BATCH_LANES streams): two chains of unequal width fill each other's launch gaps.  Measured on large-v3
    // Q5_0 (profiles/r04_stream_scaling.txt, r04_chain_split_sweep.txt): 16 states as 10 + 6: 14.2 chunks/s, 12 + 4: 13.9, 8 + 8: 11.5-12.5, one
    // chain of 16: 12.9; 32 as 20 + 12: 18.0, 16 + 16: 15.3; 8 as 5 + 3 or 6 + 2: 9.4-10.0, one chain of 8: 9.3; 6 as 4 + 2: 8.2, one chain of 6: 7.4; 7 as 5 + 2: 8.7, 6 + 1: 7.2; three or more chains
    // (40 %): 11.0 at 32; chains of 3 + 2 at 5 states: 4.9 (one chain of 4 + a solo state: 6.6).
    // Swept again with the matrix-core mat-vecs (profiles/r05_chain_split_sweep.txt; 16 / 32 states): 34 %: 8.9 / 14.4, 40 %: 10.0 / 14.3, 50 %: 11.4 / 19.3, 60 %: 15.7 / 21.1,
    // 70 %: 16.0 / 22.2, 80 %: 15.0 / 19.1, 90 %: 14.0 / 21.2.
    // More than 8 columns travel as images of 8 (mi355x_kernels.h: MI355X_IMG_COLS): the weights are still read once per chain step.
    static const int env_cols = getenv("GGML_MI355X_BATCH_COLS") ? std::max(2, std::min(MI355X_MAX_COLS, atoi(getenv("GGML_MI355X_BATCH_COLS")))) : 0;
    mi_batch_group & grp = g_batch[b->device];
    // two whisper_contexts on one device (two copies of the weights, or two different models) never share a chain: the waiting set is partitioned by
    // this signature when a leader gathers its columns (a mixed set used to fall back to one chain per state on EVERY step)
    mi_batch_member me = { b, cgraph, 0, GGML_STATUS_SUCCESS, cgraph->n_nodes, cgraph->nodes[cgraph->n_nodes - 1]->src[0]->data };
    std::unique_lock<std::mutex> lk(grp.m);
    if (!b->in_group) { b->in_group = true; grp.members.push_back(b); }
    constexpr int split_pct = 60, split_min = 4;
    // More decoding states than one chain can carry (round 6, profiles/r06_streams_over_32.txt): the 60 % rule then cuts the set into one full chain and an odd rest
    // (48 states = 29 + 19: 11.7 chunks/s, 618 closed windows) — chains of EQUAL width in the smallest possible number do better: 48 = 24 + 24: 21.8, 64 = 32 + 32: 24.3
    // (fixed widths: 64 as chains of 24: 17.7, of 16: 18.9; 48 as 32 + 16: 18.6, as 3 x 16: 21.2).
    auto cols_cap = [&]() {
        if (env_cols) return env_cols;
        const int n = (int) grp.members.size();
        if (n > MI355X_MAX_COLS) { const int k = (n + MI355X_MAX_COLS - 1) / MI355X_MAX_COLS; return (n + k - 1) / k; }
        return std::min(MI355X_MAX_COLS, std::max(split_min, (split_pct * n + 99) / 100));
    };
    if ((int) grp.members.size() < mi_batch_min_states()) {
        // too few decoding states for a merged chain to pay: this step runs on the state's own stream (it stays counted)
        bool idle = true;
        for (int i = 0; i < MI_BATCH_LANES; i++) idle = idle && !grp.lanes[i].busy;
        if (idle && grp.waiting.empty()) { grp.n_solo++; lk.unlock(); return mi_compute_own(b, cgraph); }
    }
    grp.waiting.push_back(&me);
    const double arrived = now_ms();
    grp.cv.notify_all();                                      // a waiter may now have its full set
    for (;;) {
        if (me.state == 2) return me.status;
        int lane = -1;
        for (int i = 0; i < MI_BATCH_LANES && lane < 0; i++) if (!grp.lanes[i].busy) lane = i;
        bool lead = false;
        if (me.state == 0 && lane >= 0) {
            // states that are on their way through a running chain come back later: the set to wait for is everybody else
            int in_flight = 0;
            for (int i = 0; i < MI_BATCH_LANES; i++) if (grp.lanes[i].busy) in_flight += grp.lane_cols[i];
            const int want = std::max(1, std::min<int>((int) grp.members.size() - in_flight, cols_cap()));
            if ((int) grp.waiting.size() >= want) lead = true;
            else if (grp.waiting.front() == &me && now_ms() > std::max(arrived, grp.last_finish_ms) + window_ms) {
                // the window closed: whoever is counted but neither here nor a column of a running chain is dropped (it rejoins with its
                // next step)
                for (size_t i = 0; i < grp.members.size(); ) {
                    bool here = grp.members[i]->in_flight;
                    for (auto * w : grp.waiting) here = here || w->b == grp.members[i];
                    if (!here) { grp.members[i]->in_group = false; grp.members.erase(grp.members.begin() + i); } else i++;
                }
                grp.n_timeouts++;
                lead = true;
            }
        }
        if (!lead) {
#if defined(__SANITIZE_THREAD__)
            // (the sanitizer build waits through pthread_cond_timedwait: gcc 11's libtsan has no interceptor for pthread_cond_clockwait, which wait_for uses —
            //  it would not see the mutex released inside the wait and report a double lock plus a race on everything the mutex guards)
            grp.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(200));
#else
            grp.cv.wait_for(lk, std::chrono::microseconds(200));
#endif
            continue;
        }
        mi_batch_member * mem[MI355X_MAX_COLS];
        int n = 0;
        const int max_cols = cols_cap();
        for (size_t i = 0; n < max_cols && i < grp.waiting.size(); ) {
            mi_batch_member * w = grp.waiting[i];
            if (n > 0 && (w->sig_nodes != mem[0]->sig_nodes || w->sig_w != mem[0]->sig_w)) { i++; continue; }      // another model: it leads (or joins) a chain of its own
            mem[n] = w; w->state = 1; w->b->in_flight = true; grp.waiting.erase(grp.waiting.begin() + i); n++;
        }
        mi_batch_group::lane & ln = grp.lanes[lane];
        ln.busy = true; grp.lane_cols[lane] = n;
        lk.unlock();
        bool merged = false;
        if (n == 1) { mem[0]->status = mi_compute_own(mem[0]->b, mem[0]->g); }
        else        merged = mi_compute_batch(grp, ln, mem, n);
        lk.lock();
        if (n == 1) grp.n_solo++; else if (merged) { grp.n_batches++; grp.n_columns += (uint64_t) n; }
        for (int c = 0; c < n; c++) { mem[c]->state = 2; mem[c]->b->in_flight = false; }
        ln.busy = false; grp.lane_cols[lane] = 0;
        grp.last_finish_ms = now_ms();
        grp.cv.notify_all();
    }
}

