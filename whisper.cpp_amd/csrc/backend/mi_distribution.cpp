// mi_distribution.cpp — part of the MI355X ggml backend plugin; see mi_backend.h for the map of the translation units.
#include "mi_backend.h"

// ---------------------------------------------------------------------------------------------------
// multi-GPU weight distribution: replicas are independent streams, the only exchange is the ONE-TIME copy
// of rank 0's WEIGHTS buffers into the identically laid out buffers of the other replicas (every context allocates the same
// tensors in the same order, src/whisper.cpp:1685-1859, so buffer i has the same size everywhere — checked).  Two transports:
//   * one process, several devices   : RCCL — one communicator per device (ncclCommInitAll), ONE grouped ncclBroadcast per weights
//     buffer from the first device (the reference's own in-process pattern: ggml-cuda.cu:1188 ncclCommInitAll, :1027-1030 grouped
//     collectives)                                                                           ggml_backend_mi355x_broadcast_weights_rccl_group
//     or, only when asked for, hipMemcpyPeerAsync (xGMI peer copy)                           ggml_backend_mi355x_broadcast_weights_peer
//   * one process per device (torchrun): RCCL ncclBroadcast on a communicator built here from a 128-byte unique id that the host
//     harness hands to every rank (librccl.so is dlopen()ed: the plugin does not link it)   ggml_backend_mi355x_broadcast_weights_rccl
// Either way every buffer is then check-summed on the device (mi355x_checksum) and compared with the source's: a replica that does
// not hold rank 0's bytes is an error, never a silently different model.
// ---------------------------------------------------------------------------------------------------
static std::vector<mi_weight_rec> mi_weight_list(int device) {
    std::lock_guard<std::mutex> lk(g_weights_mtx);
    std::vector<mi_weight_rec> v;
    for (auto & r : g_buffers) if (r.device == device && r.buf->usage == GGML_BACKEND_BUFFER_USAGE_WEIGHTS) v.push_back(r);
    return v;
}

static int mi_checksums(int device, const std::vector<mi_weight_rec> & bufs, std::vector<uint64_t> & sums) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    void * d = nullptr;
    if (hipMalloc(&d, 16) != hipSuccess) return -1;
    sums.assign(bufs.size() * 2, 0);
    int rc = 0;
    for (size_t i = 0; i < bufs.size() && rc == 0; i++) {
        if (mi355x_checksum(nullptr, bufs[i].base, bufs[i].size, d) != 0 || hipMemcpy(&sums[2*i], d, 16, hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
    }
    (void) hipFree(d);
    return rc;
}

#include <dlfcn.h>
#include <unistd.h>
// RCCL is dlopen()ed; only a handful of its types are needed here.  With the RCCL headers installed they come from there, on a ROCm
// install without them the same (ABI-stable, nccl.h) declarations are made locally so that the plugin still builds.
#if defined(__has_include) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm * ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
#endif

// RCCL announces itself on STDOUT when a communicator is created ("Librccl path : ..."): a plugin must not write into its host's stdout
// (bench.py's one JSON line, whisper-cli's transcript).  While a communicator is being set up, file descriptor 1 points at stderr.
struct mi_stdout_to_stderr {
    int saved = -1;
    mi_stdout_to_stderr() { fflush(stdout); saved = dup(1); if (saved >= 0) (void) dup2(2, 1); }
    ~mi_stdout_to_stderr() { if (saved >= 0) { fflush(stdout); (void) dup2(saved, 1); close(saved); } }
};

struct mi_rccl_api {
    void * h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*GroupStart)(void) = nullptr;
    ncclResult_t (*GroupEnd)(void) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char * (*GetErrorString)(ncclResult_t) = nullptr;
};
static std::string g_rccl_why;           // why no usable librccl was found (written once, under the call_once)
static mi_rccl_api * mi_rccl() {
    static mi_rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        // (a host that already loaded an RCCL — PyTorch bundles one under the same soname — gets that one back from the first name: one RCCL per process)
        for (const char * n : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" }) {
            (void) dlerror();
            void * h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) { const char * e = dlerror(); g_rccl_why += std::string(n) + ": " + (e ? e : "dlopen failed") + "; "; continue; }
            mi_rccl_api a; a.h = h;
            const char * missing = nullptr;
            auto sym = [&](const char * name) { void * p = dlsym(h, name); if (!p && !missing) missing = name; return p; };
            a.GetUniqueId    = (decltype(a.GetUniqueId))    sym("ncclGetUniqueId");
            a.CommInitRank   = (decltype(a.CommInitRank))   sym("ncclCommInitRank");
            a.CommInitAll    = (decltype(a.CommInitAll))    sym("ncclCommInitAll");
            a.GroupStart     = (decltype(a.GroupStart))     sym("ncclGroupStart");
            a.GroupEnd       = (decltype(a.GroupEnd))       sym("ncclGroupEnd");
            a.CommDestroy    = (decltype(a.CommDestroy))    sym("ncclCommDestroy");
            a.Broadcast      = (decltype(a.Broadcast))      sym("ncclBroadcast");
            a.AllReduce      = (decltype(a.AllReduce))      sym("ncclAllReduce");
            a.GetErrorString = (decltype(a.GetErrorString)) dlsym(h, "ncclGetErrorString");
            if (missing) { g_rccl_why += std::string(n) + ": no symbol " + missing + "; "; dlclose(h); continue; }
            api = a;
            break;
        }
    });
    return api.h ? &api : nullptr;
}

extern "C" {

// uploads of weight tensors are skipped while the flag is set (replicas whose weights will arrive by broadcast: the skipping
// model loader of the host harness never reads tensor payloads, this covers callers that do)
void ggml_backend_mi355x_defer_weights(int on) { t_defer_weights = on ? 1 : 0; }
uint64_t ggml_backend_mi355x_deferred_bytes(void) { return g_deferred_bytes.load(); }

// out[0..2n): {sum, weighted sum} of every WEIGHTS buffer of `device` in allocation order; returns n (or -1)
int ggml_backend_mi355x_weights_checksum(int device, uint64_t * out, int cap) {
    mi_shadows_drop(device, nullptr);
    if (hipSetDevice(device) == hipSuccess) { mi_io_drain(device); (void) hipDeviceSynchronize(); }
    const std::vector<mi_weight_rec> bufs = mi_weight_list(device);
    std::vector<uint64_t> sums;
    if (mi_checksums(device, bufs, sums) != 0) return -1;
    for (size_t i = 0; i < bufs.size() && (int) i < cap; i++) { out[2*i] = sums[2*i]; out[2*i + 1] = sums[2*i + 1]; }
    return (int) bufs.size();
}

// in-process: copy every WEIGHTS buffer of src_device into the same-index buffer of dst_device, then verify.
// stats[0..3] = bytes, seconds, buffers, verified (1/0).  Returns 0, or a negative code (-2 layout mismatch, -3 copy failed, -4 checksum mismatch).
static int mi_copy_verify(int src_device, int dst_device, const std::vector<mi_weight_rec> & S, const std::vector<mi_weight_rec> & D, double * stats) {
    if (S.empty() || S.size() != D.size()) return -2;
    for (size_t i = 0; i < S.size(); i++) if (S[i].size != D[i].size) return -2;
    for (int dev : { src_device, dst_device }) { if (hipSetDevice(dev) != hipSuccess) return -3; mi_io_drain(dev); (void) hipDeviceSynchronize(); }
    if (src_device != dst_device) {
        int can = 0;
        (void) hipDeviceCanAccessPeer(&can, dst_device, src_device);
        if (can) { (void) hipSetDevice(dst_device); hipError_t e = hipDeviceEnablePeerAccess(src_device, 0); if (e != hipSuccess) (void) hipGetLastError(); }
    }
    const double t0 = now_ms();
    double bytes = 0;
    (void) hipSetDevice(dst_device);
    for (size_t i = 0; i < S.size(); i++) {
        const hipError_t e = src_device != dst_device ? hipMemcpyPeerAsync(D[i].base, dst_device, S[i].base, src_device, S[i].size, nullptr)
                                                      : hipMemcpyAsync(D[i].base, S[i].base, S[i].size, hipMemcpyDeviceToDevice, nullptr);
        if (e != hipSuccess) return -3;
        bytes += (double) S[i].size;
    }
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    const double secs = (now_ms() - t0) * 1e-3;
    std::vector<uint64_t> cs, cd;
    if (mi_checksums(src_device, S, cs) != 0 || mi_checksums(dst_device, D, cd) != 0) return -4;
    const bool ok = cs == cd;
    if (stats) { stats[0] += bytes; stats[1] += secs; stats[2] = (double) S.size(); stats[3] = ok ? 1 : 0; }
    return ok ? 0 : -4;
}

int ggml_backend_mi355x_broadcast_weights_peer(int src_device, int dst_device, double * stats) {
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (src_device == dst_device) return -2;
    mi_shadows_drop(dst_device, nullptr);
    return mi_copy_verify(src_device, dst_device, mi_weight_list(src_device), mi_weight_list(dst_device), stats);
}

// ONE process, n devices (the form `python bench.py --gpus N` and mi355x_host_run use): one RCCL communicator per device from
// ncclCommInitAll, then per weights buffer ONE grouped ncclBroadcast (ncclGroupStart .. one call per rank .. ncclGroupEnd) from
// devices[0] into the same-index buffer of every other device, each rank on its own non-blocking stream; afterwards every buffer of every
// device is check-summed on its device and compared with devices[0]'s.  n == 1 is a valid world (communicator, broadcast and verification
// all run; nothing moves).  stats[0..5] = bytes delivered (size x (n - 1)), seconds of the broadcasts alone, buffers, verified (1/0),
// seconds of ncclCommInitAll, ranks.  Returns 0, -1 RCCL unavailable / failed, -2 layout differs between devices, -4 checksum mismatch.
int ggml_backend_mi355x_broadcast_weights_rccl_group(const int * devices, int n, double * stats) {
    if (stats) for (int i = 0; i < 6; i++) stats[i] = 0;
    mi_rccl_api * r = mi_rccl();
    if (!r) { GGML_LOG_ERROR("ggml-mi355x: no usable librccl: %s\n", g_rccl_why.c_str()); return -1; }
    if (n < 1 || n > MI_MAX_DEVICES || !devices) return -1;
    auto nccl_fail = [&](const char * what, ncclResult_t e) { GGML_LOG_ERROR("ggml-mi355x: %s failed: %s\n", what, r->GetErrorString ? r->GetErrorString(e) : "?"); };
    for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) if (devices[a] == devices[b]) return -2;      // RCCL: one rank per device
    std::vector<std::vector<mi_weight_rec>> B(n);
    for (int k = 0; k < n; k++) {
        mi_shadows_drop(devices[k], nullptr);
        if (hipSetDevice(devices[k]) != hipSuccess) return -1;
        mi_io_drain(devices[k]); (void) hipDeviceSynchronize();
        B[k] = mi_weight_list(devices[k]);
    }
    if (B[0].empty()) return -2;
    for (int k = 1; k < n; k++) {
        if (B[k].size() != B[0].size()) return -2;
        for (size_t i = 0; i < B[0].size(); i++) if (B[k][i].size != B[0][i].size) return -2;
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    std::vector<hipStream_t> st(n, nullptr);
    const double ti = now_ms();
    {
        mi_stdout_to_stderr quiet;
        const ncclResult_t e = r->CommInitAll(comms.data(), n, devices);
        if (e != ncclSuccess) { nccl_fail("ncclCommInitAll", e); return -1; }
    }
    const double init_s = (now_ms() - ti) * 1e-3;
    int rc = 0;
    for (int k = 0; k < n && rc == 0; k++) if (hipSetDevice(devices[k]) != hipSuccess || hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) != hipSuccess) rc = -1;
    double bytes = 0, secs = 0;
    if (rc == 0) {
        const double t0 = now_ms();
        for (size_t i = 0; i < B[0].size() && rc == 0; i++) {
            ncclResult_t e = r->GroupStart();
            if (e != ncclSuccess) { nccl_fail("ncclGroupStart", e); rc = -1; break; }
            for (int k = 0; k < n; k++) if ((e = r->Broadcast(B[k][i].base, B[k][i].base, B[0][i].size, ncclUint8, 0, comms[k], st[k])) != ncclSuccess) { nccl_fail("ncclBroadcast", e); rc = -1; }
            if ((e = r->GroupEnd()) != ncclSuccess) { nccl_fail("ncclGroupEnd", e); rc = -1; }
            bytes += (double) B[0][i].size * (n - 1);
        }
        for (int k = 0; k < n; k++) if (hipSetDevice(devices[k]) != hipSuccess || hipStreamSynchronize(st[k]) != hipSuccess) { GGML_LOG_ERROR("ggml-mi355x: synchronizing the broadcast stream of device %d failed\n", devices[k]); rc = -1; }
        secs = (now_ms() - t0) * 1e-3;
    }
    bool verified = false;
    if (rc == 0) {
        std::vector<uint64_t> c0;
        verified = mi_checksums(devices[0], B[0], c0) == 0;
        for (int k = 1; k < n && verified; k++) { std::vector<uint64_t> ck; verified = mi_checksums(devices[k], B[k], ck) == 0 && ck == c0; }
        if (!verified) rc = -4;
    }
    for (int k = 0; k < n; k++) {
        if (st[k]) { (void) hipSetDevice(devices[k]); (void) hipStreamDestroy(st[k]); }
        if (comms[k]) (void) r->CommDestroy(comms[k]);
    }
    if (stats) { stats[0] = bytes; stats[1] = secs; stats[2] = (double) B[0].size(); stats[3] = verified ? 1 : 0; stats[4] = init_s; stats[5] = n; }
    return rc;
}

// n_replicas contexts created one after the other on ONE device (a one-GPU machine standing in for n GPUs, so that the
// payload-skipping load -> copy -> verify -> run path can be executed on real hardware): the device's WEIGHTS buffers are
// n_replicas groups of k in allocation order; group 0 is copied into every other group and verified like a peer broadcast.
int ggml_backend_mi355x_clone_weights(int device, int n_replicas, double * stats) {
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    mi_shadows_drop(device, nullptr);
    const std::vector<mi_weight_rec> all = mi_weight_list(device);
    if (n_replicas < 2 || all.empty() || all.size() % (size_t) n_replicas) return -2;
    const size_t k = all.size() / (size_t) n_replicas;
    const std::vector<mi_weight_rec> S(all.begin(), all.begin() + k);
    for (int g = 1; g < n_replicas; g++) {
        const std::vector<mi_weight_rec> D(all.begin() + g * k, all.begin() + (g + 1) * k);
        const int rc = mi_copy_verify(device, device, S, D, stats);
        if (rc != 0) return rc;
    }
    return 0;
}

int ggml_backend_mi355x_rccl_unique_id(void * out128) {
    mi_rccl_api * r = mi_rccl();
    if (!r) return -1;
    ncclUniqueId id;
    { mi_stdout_to_stderr quiet; if (r->GetUniqueId(&id) != ncclSuccess) return -1; }
    memcpy(out128, &id, sizeof(id));
    return 0;
}

// one process per device: ncclBroadcast of every WEIGHTS buffer of `device` from rank 0 over a communicator created here from the
// shared unique id, then a checksum of every buffer compared across ALL ranks (allreduce min / max must agree).
// stats[0..3] as above.  Returns 0 or a negative code (-1 RCCL unavailable / failed, -2 layout mismatch between ranks, -4 checksum mismatch).
int ggml_backend_mi355x_broadcast_weights_rccl(int device, int rank, int world, const void * unique_id128, double * stats) {
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    mi_rccl_api * r = mi_rccl();
    if (!r || world < 1 || rank < 0 || rank >= world) return -1;
    mi_shadows_drop(device, nullptr);
    if (hipSetDevice(device) != hipSuccess) return -1;
    mi_io_drain(device); (void) hipDeviceSynchronize();
    const std::vector<mi_weight_rec> B = mi_weight_list(device);
    ncclUniqueId id; memcpy(&id, unique_id128, sizeof(id));
    // Local resources FIRST: a rank that cannot get them still joins the communicator and the agreement round below, so that no other
    // rank is left waiting inside a collective for it.  (A rank whose ncclCommInitRank itself fails cannot be helped from here: the
    // others block in their own init until RCCL's bootstrap times out — the host harness's all_ranks_ok round then reports it.)
    hipStream_t st = nullptr;
    uint64_t * d64 = nullptr;          // device scratch: layout words, then checksums
    const size_t nw = 1 + B.size();
    const size_t cap_words = 3 * std::max(nw, 2 * B.size() + 2) + 8;
    bool local_ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    if (hipMalloc((void **) &d64, cap_words * 8) != hipSuccess) { d64 = nullptr; local_ok = false; (void) hipGetLastError(); }
    ncclComm_t comm = nullptr;
    ncclResult_t init_rc;
    { mi_stdout_to_stderr quiet; init_rc = r->CommInitRank(&comm, world, id, rank); }
    if (init_rc != ncclSuccess) {
        if (d64) (void) hipFree(d64);
        if (st) (void) hipStreamDestroy(st);
        return -1;
    }
    int rc = 0;
    std::vector<uint64_t> lay(nw), lo(nw), hi(nw);
    lay[0] = B.size(); for (size_t i = 0; i < B.size(); i++) lay[1 + i] = B[i].size;
    auto all_equal = [&](std::vector<uint64_t> & v) -> bool {      // every rank holds the same words?  (min == max over ranks)
        const size_t n = v.size();
        if (hipMemcpy(d64, v.data(), n*8, hipMemcpyHostToDevice) != hipSuccess) return false;
        if (r->AllReduce(d64, d64 + n, n, ncclUint64, ncclMin, comm, st) != ncclSuccess) return false;
        if (r->AllReduce(d64, d64 + 2*n, n, ncclUint64, ncclMax, comm, st) != ncclSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        std::vector<uint64_t> a(n), b(n);
        if (hipMemcpy(a.data(), d64 + n, n*8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(b.data(), d64 + 2*n, n*8, hipMemcpyDeviceToHost) != hipSuccess) return false;
        return a == b;
    };
    if (!local_ok) {
        // this rank cannot run the data collectives: it cannot take part in the agreement either (no device word), so it leaves;
        // the communicator is destroyed, which makes the peers' next collective fail instead of hang
        if (d64) (void) hipFree(d64);
        if (st) (void) hipStreamDestroy(st);
        (void) r->CommDestroy(comm);
        return -1;
    }
    if (rc == 0 && !all_equal(lay)) rc = -2;                        // same number of buffers, same sizes, on every rank
    double bytes = 0, secs = 0;
    if (rc == 0) {
        const double t0 = now_ms();
        for (size_t i = 0; i < B.size() && rc == 0; i++) {
            if (r->Broadcast(B[i].base, B[i].base, B[i].size, ncclUint8, 0, comm, st) != ncclSuccess) rc = -1;
            bytes += (double) B[i].size;
        }
        if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = -1;
        secs = (now_ms() - t0) * 1e-3;
    }
    bool verified = false;
    if (rc != -2) {
        // every rank that agreed on the layout takes part in this round, also one whose broadcast or checksum failed locally: it
        // contributes words that cannot match (its rank in the high bits), so ALL ranks see "not verified" and nobody waits forever
        std::vector<uint64_t> cs;
        const bool local = rc == 0 && mi_checksums(device, B, cs) == 0;
        if (!local) cs.assign(2 * B.size(), 0xBAD0000000000000ull | (uint64_t) rank);
        cs.push_back(0xC0FFEEull); cs.push_back((uint64_t) B.size());
        verified = all_equal(cs) && local;
        if (!verified && rc == 0) rc = -4;
    }
    if (d64) (void) hipFree(d64);
    if (st) (void) hipStreamDestroy(st);
    (void) r->CommDestroy(comm);
    if (stats) { stats[0] = bytes; stats[1] = secs; stats[2] = (double) B.size(); stats[3] = verified ? 1 : 0; }
    return rc;
}

} // extern "C"
