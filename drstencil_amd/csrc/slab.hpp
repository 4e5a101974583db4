// slab.hpp -- the N > 1 entry points of the C ABI (include/drstencil_amd.h: drs_slab_*): one rank of a slab-decomposed run,
// native.  Included by capi.cpp (it drives drs_kernel objects).
//
// The reference is single-GPU ("Distributed communication backend: none"); what this must reproduce is the
// single-domain run, and its reference implementation is drstencil_amd/multigpu.py (SlabPlan / SlabRun, the path bench.py --gpus N
// uses through torch.distributed): the same plan (balanced z slabs in 3D / y slabs in 2D, ghost width G = every * H on interior
// faces, boundary views first, interior view, exchange of the just-written buffer's boundary planes with <= 2 neighbours), the
// same kernels (ordinary generated kernels on contiguous views; dr2_<name> for the two boundary views in one launch), bit for bit
// (tests/test_gpu_parity.py::test_native_slab_loop_equals_the_torch_one).  What is different is the plumbing:
//   * RCCL is called directly: ncclCommInitRank from an id the caller distributes, ncclGroupStart; ncclSend / ncclRecv x <= 4;
//     ncclGroupEnd on a high-priority side stream -- no Python, no batch_isend_irecv (~100 us of host time per call);
//   * one ping-pong pair (two launches + their exchange) is captured ONCE into a HIP graph per (A, B) buffer pair and replayed:
//     the host cost of a pair is one hipGraphLaunch (torch's RCCL watchdog rejects capture in the Python path; there is no
//     watchdog here).  If the capture is refused the loop runs eagerly and drs_slab_info says so.
// librccl is loaded with dlopen on first use, so libdrstencil_amd.so itself does not depend on it.
#pragma once
#include <rccl/rccl.h>

struct SlabView { long a = 0, b = 0; bool valid = false; long len() const { return b - a; } };

struct drs_slab {
    // plan (multigpu.SlabPlan)
    int ndim = 3, world = 1, rank = 0, every = 1, H = 0, G = 0;
    long L = 0, z0 = 0, z1 = 0, lo = 0, hi = 0, Lloc = 0, plane_elems = 0;
    int elem_bytes = 4, step = 1, iterations = 0;
    bool has_up = false, has_dn = false, self_neighbour = false;
    SlabView top, bot, interior, send_up, recv_up, send_dn, recv_dn;
    // kernels (owned)
    drs_kernel *k_top = nullptr, *k_bot = nullptr, *k_pair = nullptr, *k_interior = nullptr, *k_full = nullptr;
    // runtime
    ncclComm_t comm = nullptr;
    hipStream_t main = nullptr, side = nullptr;
    bool own_main = false;
    hipEvent_t ev_b = nullptr, ev_c = nullptr;
    hipGraphExec_t graph = nullptr;
    void *graph_a = nullptr, *graph_b = nullptr;
    int graph_state = 0;       // 0 not tried, 1 captured, -1 refused (eager)
    std::string info, error;
};

namespace {

struct Rccl {
    void *dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool load() {
        if (dl) return true;
        // the copy a host process already has (PyTorch ships its own librccl.so) before ROCm's
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) if ((dl = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
        if (!dl) for (const char *n : names) if ((dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!dl) { error = std::string("librccl not found: ") + dlerror(); return false; }
#define DRS_SYM(field, name) field = (decltype(field))dlsym(dl, name); if (!field) { error = std::string("librccl lacks ") + name; dl = nullptr; return false; }
        DRS_SYM(GetUniqueId, "ncclGetUniqueId") DRS_SYM(CommInitRank, "ncclCommInitRank") DRS_SYM(CommDestroy, "ncclCommDestroy")
        DRS_SYM(Send, "ncclSend") DRS_SYM(Recv, "ncclRecv") DRS_SYM(GroupStart, "ncclGroupStart") DRS_SYM(GroupEnd, "ncclGroupEnd")
        DRS_SYM(GetErrorString, "ncclGetErrorString")
#undef DRS_SYM
        return true;
    }
};
Rccl g_rccl;

// a copy of the spec with its outermost size replaced (multigpu._write_view_stc: same file names, so both paths share kernels)
std::string write_view_stc(const std::string &base_stc, int ndim, long Lv, const std::string &cache_dir) {
    std::string text = read_text(base_stc);
    const char key = ndim == 3 ? 'L' : 'M';
    size_t at = std::string::npos;
    for (size_t i = 0; i + 1 < text.size(); i++) {
        if (text[i] != key || (i > 0 && !isspace((unsigned char)text[i - 1])) || !isspace((unsigned char)text[i + 1])) continue;
        size_t j = i + 1;
        while (j < text.size() && isspace((unsigned char)text[j])) j++;
        if (j < text.size() && isdigit((unsigned char)text[j])) { at = j; break; }
    }
    if (at == std::string::npos) return "";
    size_t end = at;
    while (end < text.size() && isdigit((unsigned char)text[end])) end++;
    text.replace(at, end - at, std::to_string(Lv));
    std::string name = base_stc;
    size_t slash = name.find_last_of('/');
    if (slash != std::string::npos) name = name.substr(slash + 1);
    if (name.size() >= 4) name.erase(name.size() - 4);
    const std::string path = cache_dir + "/" + name + "_slabL" + std::to_string(Lv) + ".stc";
    char tmpl[64];
    snprintf(tmpl, sizeof tmpl, ".%d.tmp", (int)getpid());
    if (!write_text(path + tmpl, text)) return "";
    rename((path + tmpl).c_str(), path.c_str());
    return path;
}

drs_kernel *build_view_kernel(const std::vector<std::string> &opts, const std::string &stc, bool pair, const char *cache_dir, std::string &err) {
    std::vector<std::string> a = opts;
    if (pair) { a.push_back("--pair-launch"); a.push_back("1"); }
    a.push_back(stc);
    std::vector<const char *> av;
    for (auto &s : a) av.push_back(s.c_str());
    char *log = nullptr;
    drs_kernel *k = drs_kernel_build((int)av.size(), av.data(), cache_dir, &log);
    if (!k) err = log ? log : "kernel build failed";
    free(log);
    return k;
}

inline char *view_ptr(const drs_slab *s, void *buf, long plane) { return (char *)buf + (size_t)plane * s->plane_elems * s->elem_bytes; }

// planes of `dst` to the neighbours and theirs into the ghost planes: one RCCL group on the side stream
int slab_exchange(drs_slab *s, void *dst) {
    const size_t cnt = (size_t)s->G * s->plane_elems;
    const ncclDataType_t dt = s->elem_bytes == 4 ? ncclFloat : ncclDouble;
    ncclResult_t rc = g_rccl.GroupStart();
    if (s->self_neighbour) {      // rehearsal of a middle rank on one GPU (multigpu.SelfNeighbourRun): what goes "up" arrives in the lower ghosts
        if (rc == ncclSuccess) rc = g_rccl.Send(view_ptr(s, dst, s->send_up.a), cnt, dt, 0, s->comm, s->side);
        if (rc == ncclSuccess) rc = g_rccl.Recv(view_ptr(s, dst, s->recv_dn.a), cnt, dt, 0, s->comm, s->side);
        if (rc == ncclSuccess) rc = g_rccl.Send(view_ptr(s, dst, s->send_dn.a), cnt, dt, 0, s->comm, s->side);
        if (rc == ncclSuccess) rc = g_rccl.Recv(view_ptr(s, dst, s->recv_up.a), cnt, dt, 0, s->comm, s->side);
    } else {
        if (s->has_up) {
            if (rc == ncclSuccess) rc = g_rccl.Send(view_ptr(s, dst, s->send_up.a), cnt, dt, s->rank - 1, s->comm, s->side);
            if (rc == ncclSuccess) rc = g_rccl.Recv(view_ptr(s, dst, s->recv_up.a), cnt, dt, s->rank - 1, s->comm, s->side);
        }
        if (s->has_dn) {
            if (rc == ncclSuccess) rc = g_rccl.Send(view_ptr(s, dst, s->send_dn.a), cnt, dt, s->rank + 1, s->comm, s->side);
            if (rc == ncclSuccess) rc = g_rccl.Recv(view_ptr(s, dst, s->recv_dn.a), cnt, dt, s->rank + 1, s->comm, s->side);
        }
    }
    ncclResult_t rc2 = g_rccl.GroupEnd();
    if (rc == ncclSuccess) rc = rc2;
    if (rc != ncclSuccess) { s->error = std::string("RCCL: ") + g_rccl.GetErrorString(rc); return -1; }
    return 0;
}

// one launch src -> dst with the halo exchange of dst (SlabRun.launch): boundary views first, event b, interior on the main stream,
// the exchange on the side stream behind b, the main stream waits for it (event c) before the next launch reads the ghosts
int slab_launch_exchanging(drs_slab *s, void *src, void *dst) {
    const bool exch = s->world > 1 || s->self_neighbour;
    if (s->k_pair) {
        if (s->k_pair->launch_pair(view_ptr(s, src, s->top.a), view_ptr(s, dst, s->top.a), view_ptr(s, src, s->bot.a), view_ptr(s, dst, s->bot.a), s->main) != 0) return -1;
    } else {
        if (s->k_top && s->k_top->launch(view_ptr(s, src, s->top.a), view_ptr(s, dst, s->top.a), s->main) != 0) return -1;
        if (s->k_bot && s->k_bot->launch(view_ptr(s, src, s->bot.a), view_ptr(s, dst, s->bot.a), s->main) != 0) return -1;
    }
    if (exch && hipEventRecord(s->ev_b, s->main) != hipSuccess) return -1;
    if (s->k_interior && s->k_interior->launch(view_ptr(s, src, s->interior.a), view_ptr(s, dst, s->interior.a), s->main) != 0) return -1;
    if (exch) {
        if (hipStreamWaitEvent(s->side, s->ev_b, 0) != hipSuccess) return -1;
        if (slab_exchange(s, dst) != 0) return -1;
        if (hipEventRecord(s->ev_c, s->side) != hipSuccess) return -1;
        if (hipStreamWaitEvent(s->main, s->ev_c, 0) != hipSuccess) return -1;
    }
    return 0;
}

// one ping-pong pair (codegen.hpp:581-584: k(A,B); k(B,A)) on the slab
int slab_pair(drs_slab *s, void *A, void *B) {
    if (s->every == 2) {        // whole local slab, no exchange (multigpu.SlabRun.launch_local)
        if (s->k_full->launch(A, B, s->main) != 0) return -1;
    } else if (slab_launch_exchanging(s, A, B) != 0) return -1;
    return slab_launch_exchanging(s, B, A);
}

}  // namespace

extern "C" {

int drs_slab_unique_id(void *id128) {
    if (!g_rccl.load()) return -1;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return -1;
    memcpy(id128, &id, sizeof id);
    return 0;
}

drs_slab *drs_slab_open(int argc, const char *const *argv, int alone_argc, const char *const *alone_argv, int world, int rank, int every,
                        int rehearse_world, const char *cache_dir, char **log) {
    if (log) *log = nullptr;
    auto fail = [&](const std::string &m) { if (log) *log = dup_cstr(m + "\n"); return (drs_slab *)nullptr; };
    if (argc < 1 || world < 1 || rank < 0 || (every != 1 && every != 2)) return fail("drs_slab_open: bad arguments");
    std::vector<std::string> args = to_args(argc, argv);
    const std::string base_stc = args.back();
    args.pop_back();
    GenResult r = generate(to_args(argc, argv));
    if (!r.emitted) return fail(r.messages + (r.plan.error.empty() ? "" : ("drstencil: " + r.plan.error)));
    drs_slab *s = new drs_slab();
    s->ndim = r.st.ndim;
    s->L = s->ndim == 3 ? r.st.L : r.st.M;
    s->H = r.st.halo; s->step = r.st.step; s->iterations = r.st.iterations;
    s->elem_bytes = r.plan.fp32 ? 4 : 8;
    s->plane_elems = s->ndim == 3 ? (long)r.st.M * r.st.N : (long)r.st.N;
    // rehearse_world > 0: this process plays rank `rank` of `rehearse_world` on ONE GPU, its neighbours being itself
    s->self_neighbour = rehearse_world > 0;
    const int pworld = s->self_neighbour ? rehearse_world : world;
    s->world = world; s->rank = rank;
    s->every = pworld > 1 ? every : 1;
    const long G = s->G = (long)s->every * s->H, H = s->H;
    s->z0 = (long)rank * s->L / pworld; s->z1 = (long)(rank + 1) * s->L / pworld;
    s->has_up = rank > 0; s->has_dn = rank < pworld - 1;
    if (s->self_neighbour && !(s->has_up && s->has_dn)) { delete s; return fail("drs_slab_open: rehearse a middle rank"); }
    s->lo = s->z0 - (s->has_up ? G : 0); s->hi = s->z1 + (s->has_dn ? G : 0);
    s->Lloc = s->hi - s->lo;
    if (pworld > 1 && s->z1 - s->z0 < 2 * G) { delete s; return fail("drs_slab_open: slab thinner than twice the ghost width"); }
    auto view = [](long a, long b) { SlabView v; v.a = a; v.b = b; v.valid = true; return v; };
    if (s->has_up) { s->top = view(G - H, 2 * G + H); s->send_up = view(G, 2 * G); s->recv_up = view(0, G); }
    if (s->has_dn) { s->bot = view(s->Lloc - 2 * G - H, s->Lloc - G + H); s->send_dn = view(s->Lloc - 2 * G, s->Lloc - G); s->recv_dn = view(s->Lloc - G, s->Lloc); }
    s->interior = view(s->has_up ? 2 * G - H : 0, s->has_dn ? s->Lloc - 2 * G + H : s->Lloc);
    // kernels: every view is a contiguous sub-range of the slab, i.e. an ordinary generated kernel for its length
    const std::string here = self_dir();
    const std::string cdir = cache_dir ? cache_dir : (getenv("DRS_KCACHE") ? getenv("DRS_KCACHE") : here + "/_kcache");
    mkdir(cdir.c_str(), 0777);
    std::string err;
    auto kernel_for = [&](long Lv, bool pair, const std::vector<std::string> &opts) -> drs_kernel * {
        const std::string stc = write_view_stc(base_stc, s->ndim, Lv, cdir);
        if (stc.empty()) { err = "cannot write the view spec for " + std::to_string(Lv) + " planes"; return nullptr; }
        return build_view_kernel(opts, stc, pair, cache_dir, err);
    };
    bool ok = true;
    if (s->top.valid && s->bot.valid && s->top.len() == s->bot.len()) ok = (s->k_pair = kernel_for(s->top.len(), true, args)) != nullptr;
    else {
        if (s->top.valid && s->top.len() > 2 * H) ok = ok && (s->k_top = kernel_for(s->top.len(), false, args)) != nullptr;
        if (s->bot.valid && s->bot.len() > 2 * H) ok = ok && (s->k_bot = kernel_for(s->bot.len(), false, args)) != nullptr;
    }
    if (ok && s->interior.len() > 2 * H) ok = (s->k_interior = kernel_for(s->interior.len(), false, args)) != nullptr;
    if (ok && s->every == 2) {
        std::vector<std::string> alone = alone_argc > 0 ? to_args(alone_argc, alone_argv) : args;     // launches with the GPU to themselves
        ok = (s->k_full = kernel_for(s->Lloc, false, alone)) != nullptr;
    }
    if (!ok) { std::string m = "drs_slab_open: " + err; drs_slab_close(s); return fail(m); }
    return s;
}

void drs_slab_plan(const drs_slab *s, long out[8]) {
    out[0] = s->lo; out[1] = s->hi; out[2] = s->z0; out[3] = s->z1; out[4] = s->Lloc; out[5] = s->G; out[6] = s->H; out[7] = s->every;
}

int drs_slab_connect(drs_slab *s, const void *id128, void *main_stream) {
    if (!g_rccl.load()) { s->error = g_rccl.error; return -1; }
    g_launched = true;      // HIP is up from here on: no compiler may be started by this process any more
    if (main_stream) s->main = (hipStream_t)main_stream;
    else { if (hipStreamCreateWithFlags(&s->main, hipStreamNonBlocking) != hipSuccess) { s->error = "hipStreamCreate failed"; return -1; } s->own_main = true; }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // hi is the numerically lowest = highest priority
    if (hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, hi) != hipSuccess) { s->error = "hipStreamCreateWithPriority failed"; return -1; }
    if (hipEventCreateWithFlags(&s->ev_b, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->ev_c, hipEventDisableTiming) != hipSuccess) { s->error = "hipEventCreate failed"; return -1; }
    if (s->world > 1 || s->self_neighbour) {
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        ncclResult_t rc = g_rccl.CommInitRank(&s->comm, s->self_neighbour ? 1 : s->world, id, s->self_neighbour ? 0 : s->rank);
        if (rc != ncclSuccess) { s->error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc); return -1; }
    }
    return 0;
}

// The reference's loop (codegen.hpp:581-584) on the slab; result in A.  Asynchronous on the slab's main stream.  Returns the
// number of launches (two per pair, whatever the number of kernels each is made of), -1 (drs_slab_error), or -3 when a view kernel is a
// temporal pipeline (not forced) and `iterations` exceeds its tolerance horizon (the rule of drs_kernel_run).
int drs_slab_run(drs_slab *s, void *d_a, void *d_b, int iterations) {
    const int it = iterations >= 0 ? iterations : s->iterations;
    // like drs_kernel_run: a view kernel that is an on-chip temporal pipeline keeps the tolerance up to its horizon only
    for (drs_kernel *k : {s->k_top, s->k_bot, s->k_pair, s->k_interior, s->k_full})
        if (k && k->horizon >= 0 && !k->forced && it > k->horizon) {
            s->error = "iterations " + std::to_string(it) + " exceed the tolerance horizon (" + std::to_string(k->horizon) + ") of the slab's temporal kernels: open the slab without --temporal (or with --temporal force)";
            return -3;
        }
    const bool want_graph = !(getenv("DRS_SLAB_GRAPH") && getenv("DRS_SLAB_GRAPH")[0] == '0');
    if (want_graph && (s->graph_state == 0 || (s->graph_state == 1 && (s->graph_a != d_a || s->graph_b != d_b)))) {
        // capture one pair for these buffers: the side stream joins the capture through event b and leaves it through event c
        if (s->graph) { (void)hipGraphExecDestroy(s->graph); s->graph = nullptr; }
        hipGraph_t g = nullptr;
        s->graph_state = -1;
        if (hipStreamBeginCapture(s->main, hipStreamCaptureModeRelaxed) == hipSuccess) {
            const int prc = slab_pair(s, d_a, d_b);
            const hipError_t e = hipStreamEndCapture(s->main, &g);
            if (prc == 0 && e == hipSuccess && g && hipGraphInstantiate(&s->graph, g, nullptr, nullptr, 0) == hipSuccess) {
                s->graph_state = 1; s->graph_a = d_a; s->graph_b = d_b;
            }
            if (g) (void)hipGraphDestroy(g);
        }
        (void)hipGetLastError();
        if (s->graph_state != 1) s->error = "";       // a refused capture is not an error: the loop runs eagerly
    }
    int n = 0;
    for (int t = 0; t < it; t += 2 * s->step) {
        if (want_graph && s->graph_state == 1) { if (hipGraphLaunch(s->graph, s->main) != hipSuccess) { s->error = "hipGraphLaunch failed"; return -1; } }
        else if (slab_pair(s, d_a, d_b) != 0) { if (s->error.empty()) s->error = "launch failed"; return -1; }
        n += 2;
    }
    return n;
}

int drs_slab_sync(drs_slab *s) { return hipStreamSynchronize(s->main) == hipSuccess ? 0 : -1; }
void *drs_slab_stream(drs_slab *s) { return (void *)s->main; }
const char *drs_slab_error(const drs_slab *s) { return s->error.c_str(); }

const char *drs_slab_info(drs_slab *s) {
    char b[768];
    auto nm = [](drs_kernel *k) { return k ? k->path.substr(k->path.find_last_of('/') + 1) : std::string(""); };
    snprintf(b, sizeof b, "{\"world\": %d, \"rank\": %d, \"every\": %d, \"H\": %d, \"G\": %d, \"lo\": %ld, \"hi\": %ld, \"z0\": %ld, \"z1\": %ld, \"Lloc\": %ld, "
                          "\"self_neighbour\": %d, \"graph\": %d, \"kernel_pair\": \"%s\", \"kernel_interior\": \"%s\", \"kernel_full\": \"%s\"}",
             s->world, s->rank, s->every, s->H, s->G, s->lo, s->hi, s->z0, s->z1, s->Lloc, s->self_neighbour ? 1 : 0, s->graph_state,
             nm(s->k_pair).c_str(), nm(s->k_interior).c_str(), nm(s->k_full).c_str());
    s->info = b;
    return s->info.c_str();
}

void drs_slab_close(drs_slab *s) {
    if (!s) return;
    if (s->main) (void)hipStreamSynchronize(s->main);
    if (s->side) (void)hipStreamSynchronize(s->side);
    if (s->graph) (void)hipGraphExecDestroy(s->graph);
    if (s->comm && g_rccl.dl) (void)g_rccl.CommDestroy(s->comm);
    if (s->ev_b) (void)hipEventDestroy(s->ev_b);
    if (s->ev_c) (void)hipEventDestroy(s->ev_c);
    if (s->side) (void)hipStreamDestroy(s->side);
    if (s->own_main && s->main) (void)hipStreamDestroy(s->main);
    for (drs_kernel *k : {s->k_top, s->k_bot, s->k_pair, s->k_interior, s->k_full}) drs_kernel_close(k);
    delete s;
}

}  // extern "C"
