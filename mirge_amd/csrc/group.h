// group.h -- multi-GPU inside the library (BASELINE.json north_star: "reads shard across the 8 GPUs ...
// all-reduced with RCCL"), and the batch ingest of many sample files.  Included at the end of mirge_hip.hip (same
// translation unit: it drives the contexts through their internals).
//
// One mirge_group = one context per rank (per GPU), all in this process, driven by host threads.  A run is
//   collapse (every rank takes whole samples, or a contiguous chunk of a sample's reads)
//   -> merge of the unique tables by key owner: ONE all-to-all of (key | sample | count) records per key class --
//      grouped ncclSend / ncclRecv, GPU to GPU over xGMI -- and the verbatim uniques through host memory
//   -> the cascade on every rank over the sequences it now owns (each sequence exactly once in the whole job)
//   -> ONE ncclAllReduce(ncclUint64, ncclSum) of the flat count buffer, after which every rank holds the job's count vectors.
// The reference has no counterpart: it is one process with `--threads` forwarded to bowtie (miRge.pl:327).
//
// RCCL is loaded with dlopen when the first group opens (librccl.so.1: this image's /opt/rocm, or the copy a host
// program such as PyTorch has already mapped) -- the library itself does not link it, so it loads where RCCL is absent.
// MIRGE_GROUP_HOST_STAGED (flag of mirge_group_open_ex): the two exchanges go through host memory instead -- several
// ranks may then share ONE device (RCCL refuses that), which is how the N > 1 code is rehearsed on a one-GPU box.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
std::mutex g_rccl_mu;
Rccl g_rccl;
// "" = loaded; else why not
std::string rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return "";
    void* h = nullptr;
    for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { const char* why = dlerror(); return std::string("cannot load RCCL: ") + (why ? why : "librccl.so.1 not found"); }      // (dlerror() clears itself: one call)
    Rccl r; r.h = h;
#define MG_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); if (!r.field) { dlclose(h); return std::string("RCCL lacks ") + name; }
    MG_SYM(CommInitAll, "ncclCommInitAll") MG_SYM(CommDestroy, "ncclCommDestroy") MG_SYM(GroupStart, "ncclGroupStart") MG_SYM(GroupEnd, "ncclGroupEnd")
    MG_SYM(Send, "ncclSend") MG_SYM(Recv, "ncclRecv") MG_SYM(AllReduce, "ncclAllReduce") MG_SYM(GetErrorString, "ncclGetErrorString")
#undef MG_SYM
    g_rccl = r;
    return "";
}

// a whole file, inflated if it is gzip (magic 1f 8b; every member of a multi-member file)
int load_file_bytes(const char* path, std::vector<uint8_t>& out, std::string& err) {
    out.clear();
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return MIRGE_E_IO; }
    unsigned char mg2[2] = {0, 0};
    const size_t got2 = fread(mg2, 1, 2, f);
    struct stat sb; const bool st_ok = fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode);
    if (!st_ok) { fclose(f); err = std::string(path) + " is not a regular file"; return MIRGE_E_IO; }
    const uint64_t fsize = (uint64_t)sb.st_size;
    if (got2 == 2 && mg2[0] == 0x1f && mg2[1] == 0x8b) {
        uint64_t hint = fsize * 4;
        unsigned char t[4];
        if (fsize >= 18 && fseek(f, -4, SEEK_END) == 0 && fread(t, 1, 4, f) == 4) {
            const uint64_t isz = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
            if (isz > hint) hint = isz;
        }
        fclose(f);
        gzFile g = gzopen(path, "rb");
        if (!g) { err = std::string("cannot open ") + path; return MIRGE_E_IO; }
        (void)gzbuffer(g, 1u << 20);
        out.resize((size_t)hint + (1u << 20));
        size_t n = 0;
        for (;;) {
            if (out.size() - n < (1u << 20)) out.resize(out.size() + out.size() / 2);
            const int r = gzread(g, out.data() + n, (unsigned)std::min<size_t>(out.size() - n, 1u << 30));
            if (r < 0) { int en = 0; const char* m = gzerror(g, &en); err = std::string(path) + ": " + (m ? m : "gzip error"); gzclose(g); return MIRGE_E_IO; }
            if (r == 0) break;
            n += (size_t)r;
        }
        int en = 0; (void)gzerror(g, &en);
        gzclose(g);
        if (en != Z_OK && en != Z_STREAM_END) { err = std::string(path) + ": corrupt or truncated gzip stream"; return MIRGE_E_IO; }
        out.resize(n);
        return MIRGE_OK;
    }
    out.resize((size_t)fsize);
    rewind(f);
    const size_t got = fsize ? fread(out.data(), 1, (size_t)fsize, f) : 0;
    fclose(f);
    if (got != fsize) { err = std::string("short read on ") + path; return MIRGE_E_IO; }
    return MIRGE_OK;
}

// Files idx[0..m) of a batch -> context c, in that order: up to `nthr` host threads read / inflate files AHEAD (a gzip
// stream inflates serially, but different files do not wait for each other) while this thread uploads and collapses the
// ones that are ready.  At most `nthr + 2` files are held (in host memory while they inflate, then in HBM) at once.
int collapse_files(mirge_ctx* c, const std::vector<int>& idx, const char* const* paths, const int32_t* samples, const char* adapter,
                   int32_t trim5, int32_t trim3, uint64_t* total_reads, uint64_t* trimmed_reads, int32_t* phred, int nthr) {
    const size_t m = idx.size();
    if (m == 0) return MIRGE_OK;
    TrimPlan tp0;
    if (adapter) { int rc = make_trim_plan(c, adapter, trim5, trim3, tp0); if (rc) return rc; }
    nthr = (int)std::max<size_t>(1, std::min<size_t>((size_t)nthr, m));
    std::vector<std::vector<uint8_t>> bufs(m);
    std::vector<void*> dbuf(m, nullptr); std::vector<uint64_t> dsize(m, 0); std::vector<uint8_t> dlast(m, (uint8_t)'\n');      // the files in HBM
    std::vector<int> state(m, 0);                                  // 0 = not loaded, 1 = ready, -1 = failed
    std::vector<std::string> errs(m);
    std::mutex mu; std::condition_variable cv;
    size_t consumed = 0; bool stop = false;
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= m) return;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || i < consumed + (size_t)nthr + 2; }); if (stop) return; }
            std::string e;
            int rc = load_file_bytes(paths[idx[i]], bufs[i], e);
            if (!rc) {
                // the worker uploads its file itself (a buffer of its own in HBM): the collapsing thread is left with the
                // kernels alone, and the uploads of different files overlap the inflating of others
                const uint64_t size = bufs[i].size();
                hipError_t he = hipSetDevice(c->device);
                if (he == hipSuccess) he = hipMalloc(&dbuf[i], size + 256);
                if (he == hipSuccess && size) he = hipMemcpy(dbuf[i], bufs[i].data(), size, hipMemcpyHostToDevice);
                if (he != hipSuccess) { rc = MIRGE_E_HIP; e = std::string("FASTQ upload: ") + hipGetErrorString(he); }
                dsize[i] = size; dlast[i] = size ? bufs[i][size - 1] : (uint8_t)'\n';
                std::vector<uint8_t>().swap(bufs[i]);
            }
            { std::lock_guard<std::mutex> lk(mu); state[i] = rc ? -1 : 1; errs[i] = e; }
            cv.notify_all();
        }
    };
    std::vector<std::thread> thr;
    for (int t = 0; t < nthr; ++t) thr.emplace_back(worker);
    int rc = MIRGE_OK;
    if (hipSetDevice(c->device) != hipSuccess) rc = fail(c, MIRGE_E_HIP, "hipSetDevice failed");
    for (size_t i = 0; i < m && rc == MIRGE_OK; ++i) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return state[i] != 0; }); }
        if (state[i] < 0) { rc = fail(c, errs[i].compare(0, 12, "FASTQ upload") == 0 ? MIRGE_E_HIP : MIRGE_E_IO, errs[i]); break; }
        const int j = idx[i];
        const uint64_t size = dsize[i];
        const uint8_t last = dlast[i];
        const uint8_t* d = (const uint8_t*)dbuf[i];
        uint64_t n = 0;
        if (adapter) {
            TrimPlan tp = tp0;
            rc = collapse_fastq_dev(c, d, size, last, samples[j], &n, &tp);
            if (trimmed_reads) trimmed_reads[j] = tp.kept;
            if (phred) phred[j] = tp.phred_reported;
        } else {
            rc = collapse_fastq_dev(c, d, size, last, samples[j], &n);
            if (trimmed_reads) trimmed_reads[j] = n;
            if (phred) phred[j] = 33;
        }
        if (total_reads) total_reads[j] = n;
        (void)hipFree(dbuf[i]); dbuf[i] = nullptr;
        { std::lock_guard<std::mutex> lk(mu); consumed = i + 1; }
        cv.notify_all();
    }
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (auto& t : thr) t.join();
    (void)hipSetDevice(c->device);
    for (auto& p : dbuf) if (p) (void)hipFree(p);
    return rc;
}

}  // namespace

struct mirge_group {
    int world = 0, S = 1;
    bool staged = false;                          // exchanges through host memory (rehearsal: ranks may share a device)
    std::vector<mirge_ctx*> ctx;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;
    std::vector<DevBuf> send16, send48, recv16, recv48, flat;
    bool merged = false, annotated = false;
    double merge_ms = 0, allreduce_ms = 0;
    std::string err;
};

namespace {

int gfail(mirge_group* g, int code, const std::string& msg) { if (g) g->err = msg; else g_open_error = msg; return code; }
// fn(rank) on one host thread per rank; the first failure is the group's error
template <class F> int group_parallel(mirge_group* g, F fn) {
    std::vector<int> rc((size_t)g->world, MIRGE_OK);
    if (g->world == 1) rc[0] = fn(0);
    else {
        std::vector<std::thread> thr;
        for (int r = 0; r < g->world; ++r) thr.emplace_back([&, r] { rc[(size_t)r] = fn(r); });
        for (auto& t : thr) t.join();
    }
    for (int r = 0; r < g->world; ++r) if (rc[(size_t)r]) return gfail(g, rc[(size_t)r], "rank " + std::to_string(r) + ": " + g->ctx[(size_t)r]->err);
    return MIRGE_OK;
}
#define NCCLCHK(g, call)                                                                                 \
    do { ncclResult_t r__ = (call);                                                                      \
         if (r__ != ncclSuccess) return gfail((g), MIRGE_E_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r__)); } while (0)

// The merge of the unique tables by key owner (mirge_amd/shard.py's merge_uniques, inside the library).
int group_merge(mirge_group* g) {
    const int W = g->world;
    auto t0 = std::chrono::steady_clock::now();
    // what every rank holds for every owner
    std::vector<std::vector<uint64_t>> n16((size_t)W, std::vector<uint64_t>((size_t)W)), n48 = n16;
    int rc = group_parallel(g, [&](int r) {
        mirge_ctx* c = g->ctx[(size_t)r];
        if (hipSetDevice(c->device) != hipSuccess) return fail(c, MIRGE_E_HIP, "hipSetDevice failed");
        return shard_export_sizes(c, W, n16[(size_t)r].data(), n48[(size_t)r].data());
    });
    if (rc) return rc;
    std::vector<uint64_t> s16((size_t)W, 0), s48((size_t)W, 0), r16((size_t)W, 0), r48((size_t)W, 0);
    for (int r = 0; r < W; ++r) for (int d = 0; d < W; ++d) { s16[(size_t)r] += n16[(size_t)r][(size_t)d]; s48[(size_t)r] += n48[(size_t)r][(size_t)d]; r16[(size_t)d] += n16[(size_t)r][(size_t)d]; r48[(size_t)d] += n48[(size_t)r][(size_t)d]; }
    rc = group_parallel(g, [&](int r) {
        mirge_ctx* c = g->ctx[(size_t)r];
        if (hipSetDevice(c->device) != hipSuccess) return fail(c, MIRGE_E_HIP, "hipSetDevice failed");
        HIPCHK(c, g->send16[(size_t)r].ensure(std::max<uint64_t>(s16[(size_t)r], 1) * (size_t)24, false, 0, c->st));
        HIPCHK(c, g->send48[(size_t)r].ensure(std::max<uint64_t>(s48[(size_t)r], 1) * (size_t)56, false, 0, c->st));
        HIPCHK(c, g->recv16[(size_t)r].ensure(std::max<uint64_t>(r16[(size_t)r], 1) * (size_t)24, false, 0, c->st));
        HIPCHK(c, g->recv48[(size_t)r].ensure(std::max<uint64_t>(r48[(size_t)r], 1) * (size_t)56, false, 0, c->st));
        return shard_export(c, W, g->send16[(size_t)r].p, g->send48[(size_t)r].p);      // (waits for its kernels)
    });
    if (rc) return rc;
    // the all-to-all: rank r's records for owner d land in d's buffer behind those of ranks < r
    auto exchange = [&](std::vector<DevBuf>& snd, std::vector<DevBuf>& rcv, const std::vector<std::vector<uint64_t>>& n, size_t rec) -> int {
        std::vector<uint64_t> roff((size_t)W, 0);                 // per destination: bytes received so far (sources in rank order)
        if (g->staged) {
            for (int r = 0; r < W; ++r) {
                uint64_t soff = 0;
                for (int d = 0; d < W; ++d) {
                    const uint64_t bytes = n[(size_t)r][(size_t)d] * rec;
                    if (bytes) {
                        std::vector<uint8_t> h((size_t)bytes);
                        if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess || hipMemcpy(h.data(), snd[(size_t)r].as<uint8_t>() + soff, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
                            hipSetDevice(g->dev[(size_t)d]) != hipSuccess || hipMemcpy(rcv[(size_t)d].as<uint8_t>() + roff[(size_t)d], h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
                            return gfail(g, MIRGE_E_HIP, "host-staged exchange failed");
                    }
                    soff += bytes; roff[(size_t)d] += bytes;
                }
            }
            return MIRGE_OK;
        }
        // where rank r's block sits in destination d's buffer
        std::vector<std::vector<uint64_t>> at((size_t)W, std::vector<uint64_t>((size_t)W, 0));
        for (int r = 0; r < W; ++r) for (int d = 0; d < W; ++d) { at[(size_t)r][(size_t)d] = roff[(size_t)d]; roff[(size_t)d] += n[(size_t)r][(size_t)d] * rec; }
        NCCLCHK(g, g_rccl.GroupStart());
        ncclResult_t bad = ncclSuccess;                               // (a failed call must not leave the group open: ncclGroupEnd runs whatever happens)
        for (int r = 0; r < W && bad == ncclSuccess; ++r) {
            uint64_t soff = 0;
            for (int d = 0; d < W && bad == ncclSuccess; ++d) {
                const uint64_t sb = n[(size_t)r][(size_t)d] * rec, rb = n[(size_t)d][(size_t)r] * rec;
                if (sb) bad = g_rccl.Send(snd[(size_t)r].as<uint8_t>() + soff, (size_t)sb, ncclUint8, d, g->comm[(size_t)r], g->ctx[(size_t)r]->st);
                if (rb && bad == ncclSuccess) bad = g_rccl.Recv(rcv[(size_t)r].as<uint8_t>() + at[(size_t)d][(size_t)r], (size_t)rb, ncclUint8, d, g->comm[(size_t)r], g->ctx[(size_t)r]->st);
                soff += sb;
            }
        }
        const ncclResult_t ended = g_rccl.GroupEnd();
        if (bad != ncclSuccess) return gfail(g, MIRGE_E_HIP, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(bad));
        NCCLCHK(g, ended);
        for (int r = 0; r < W; ++r)
            if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess || hipStreamSynchronize(g->ctx[(size_t)r]->st) != hipSuccess) return gfail(g, MIRGE_E_HIP, "all-to-all failed");
        return MIRGE_OK;
    };
    rc = exchange(g->send16, g->recv16, n16, (size_t)24);
    if (!rc) rc = exchange(g->send48, g->recv48, n48, (size_t)56);
    if (rc) return rc;
    // verbatim uniques (a host table on every rank): every rank's blob to every rank, which keeps what it owns
    std::vector<std::vector<uint8_t>> blobs((size_t)W);
    bool any = false;
    for (int r = 0; r < W; ++r) {
        blobs[(size_t)r].resize((size_t)mirge_verb_export_size(g->ctx[(size_t)r]));
        any = any || !blobs[(size_t)r].empty();
        if (!blobs[(size_t)r].empty() && mirge_verb_export(g->ctx[(size_t)r], blobs[(size_t)r].data())) return gfail(g, MIRGE_E_ARG, g->ctx[(size_t)r]->err);
    }
    rc = group_parallel(g, [&](int r) {
        mirge_ctx* c = g->ctx[(size_t)r];
        if (hipSetDevice(c->device) != hipSuccess) return fail(c, MIRGE_E_HIP, "hipSetDevice failed");
        int q = shard_import(c, g->recv16[(size_t)r].p, r16[(size_t)r], g->recv48[(size_t)r].p, r48[(size_t)r]);
        if (q || !any) return q;
        q = mirge_verb_clear(c);
        for (int s = 0; s < W && !q; ++s) if (!blobs[(size_t)s].empty()) q = mirge_verb_import(c, blobs[(size_t)s].data(), blobs[(size_t)s].size(), r, W);
        return q;
    });
    g->merge_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!rc) g->merged = true;
    return rc;
}

// ONE sum all-reduce of the flat count buffer [quant | iscan | category | len_hist | trimmed_uniq | round statistics]
int group_allreduce(mirge_group* g) {
    const int W = g->world;
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t len = mirge_counts_flat_len(g->ctx[0]);
    int rc = group_parallel(g, [&](int r) {
        mirge_ctx* c = g->ctx[(size_t)r];
        if (hipSetDevice(c->device) != hipSuccess) return fail(c, MIRGE_E_HIP, "hipSetDevice failed");
        HIPCHK(c, g->flat[(size_t)r].ensure(len * 8, false, 0, c->st));
        return (int)mirge_counts_flat_get_dev(c, g->flat[(size_t)r].p);
    });
    if (rc) return rc;
    if (g->staged) {
        std::vector<uint64_t> sum((size_t)len, 0), h((size_t)len);
        for (int r = 0; r < W; ++r) {
            if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess || hipMemcpy(h.data(), g->flat[(size_t)r].p, len * 8, hipMemcpyDeviceToHost) != hipSuccess) return gfail(g, MIRGE_E_HIP, "host-staged all-reduce failed");
            for (uint64_t i = 0; i < len; ++i) sum[(size_t)i] += h[(size_t)i];
        }
        for (int r = 0; r < W; ++r)
            if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess || hipMemcpy(g->flat[(size_t)r].p, sum.data(), len * 8, hipMemcpyHostToDevice) != hipSuccess) return gfail(g, MIRGE_E_HIP, "host-staged all-reduce failed");
    } else {
        NCCLCHK(g, g_rccl.GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int r = 0; r < W && bad == ncclSuccess; ++r) bad = g_rccl.AllReduce(g->flat[(size_t)r].p, g->flat[(size_t)r].p, (size_t)len, ncclUint64, ncclSum, g->comm[(size_t)r], g->ctx[(size_t)r]->st);
        const ncclResult_t ended = g_rccl.GroupEnd();
        if (bad != ncclSuccess) return gfail(g, MIRGE_E_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(bad));
        NCCLCHK(g, ended);
        for (int r = 0; r < W; ++r)
            if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess || hipStreamSynchronize(g->ctx[(size_t)r]->st) != hipSuccess) return gfail(g, MIRGE_E_HIP, "all-reduce failed");
    }
    rc = group_parallel(g, [&](int r) {
        mirge_ctx* c = g->ctx[(size_t)r];
        if (hipSetDevice(c->device) != hipSuccess) return fail(c, MIRGE_E_HIP, "hipSetDevice failed");
        return (int)mirge_counts_flat_set_dev(c, g->flat[(size_t)r].p);
    });
    g->allreduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

}  // namespace

extern "C" {

const char* mirge_group_last_error(const mirge_group* g) { return g ? g->err.c_str() : g_open_error.c_str(); }

void mirge_group_close(mirge_group* g) {
    if (!g) return;
    for (int r = 0; r < (int)g->ctx.size(); ++r) {
        if (g->ctx[(size_t)r]) (void)hipSetDevice(g->dev[(size_t)r]);
        for (auto* v : {&g->send16, &g->send48, &g->recv16, &g->recv48, &g->flat}) if ((size_t)r < v->size()) (*v)[(size_t)r].release();
    }
    for (auto& cm : g->comm) if (cm) (void)g_rccl.CommDestroy(cm);
    for (auto* c : g->ctx) mirge_close(c);
    delete g;
}

mirge_group* mirge_group_open_ex(const char* species_dir, int32_t n_samples, int32_t n_ranks, const int32_t* devices, uint32_t flags) {
    if (n_ranks < 1 || n_ranks > SHARD_MAX_WORLD) { gfail(nullptr, MIRGE_E_ARG, "between 1 and " + std::to_string(SHARD_MAX_WORLD) + " ranks"); return nullptr; }
    mirge_group* g = new mirge_group();
    g->world = n_ranks; g->S = n_samples; g->staged = (flags & MIRGE_GROUP_HOST_STAGED) != 0;
    g->ctx.assign((size_t)n_ranks, nullptr); g->dev.resize((size_t)n_ranks);
    g->send16.resize((size_t)n_ranks); g->send48.resize((size_t)n_ranks); g->recv16.resize((size_t)n_ranks); g->recv48.resize((size_t)n_ranks); g->flat.resize((size_t)n_ranks);
    for (int r = 0; r < n_ranks; ++r) g->dev[(size_t)r] = devices ? devices[r] : r;
    if (!g->staged) {
        for (int r = 0; r < n_ranks; ++r) for (int q = 0; q < r; ++q)
            if (g->dev[(size_t)r] == g->dev[(size_t)q]) { gfail(nullptr, MIRGE_E_ARG, "two ranks on one device: RCCL refuses that (MIRGE_GROUP_HOST_STAGED rehearses it)"); mirge_group_close(g); return nullptr; }
        const std::string why = rccl_load();
        if (!why.empty()) { gfail(nullptr, MIRGE_E_HIP, why); mirge_group_close(g); return nullptr; }
    }
    // one context per rank, each loading (and indexing) the libraries on its own device, side by side
    std::vector<std::string> errs((size_t)n_ranks);
    {
        std::vector<std::thread> thr;
        for (int r = 0; r < n_ranks; ++r) thr.emplace_back([&, r] {
            g->ctx[(size_t)r] = mirge_open(species_dir, n_samples, g->dev[(size_t)r]);
            if (!g->ctx[(size_t)r]) errs[(size_t)r] = mirge_last_error(nullptr);       // (thread-local: read on the thread that failed)
        });
        for (auto& t : thr) t.join();
    }
    for (int r = 0; r < n_ranks; ++r) if (!g->ctx[(size_t)r]) { gfail(nullptr, MIRGE_E_HIP, "rank " + std::to_string(r) + ": " + errs[(size_t)r]); mirge_group_close(g); return nullptr; }
    if (!g->staged) {
        g->comm.assign((size_t)n_ranks, nullptr);
        ncclResult_t nr = g_rccl.CommInitAll(g->comm.data(), n_ranks, g->dev.data());
        if (nr != ncclSuccess) { gfail(nullptr, MIRGE_E_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr)); g->comm.clear(); mirge_group_close(g); return nullptr; }
    }
    return g;
}
mirge_group* mirge_group_open(const char* species_dir, int32_t n_samples, int32_t n_gpus) { return mirge_group_open_ex(species_dir, n_samples, n_gpus, nullptr, 0u); }

int32_t mirge_group_size(const mirge_group* g) { return g ? g->world : 0; }
mirge_ctx* mirge_group_ctx(mirge_group* g, int32_t rank) { return (g && rank >= 0 && rank < g->world) ? g->ctx[(size_t)rank] : nullptr; }

int32_t mirge_group_reset(mirge_group* g, int32_t n_samples) {
    if (!g) return MIRGE_E_ARG;
    g->S = n_samples; g->merged = false; g->annotated = false;
    return group_parallel(g, [&](int r) { return (int)mirge_reset(g->ctx[(size_t)r], n_samples); });
}

int32_t mirge_group_load_library_mem(mirge_group* g, int32_t lib, int32_t n, const char* names, const char* seqs, const uint64_t* offs) {
    if (!g) return MIRGE_E_ARG;
    return group_parallel(g, [&](int r) { return (int)mirge_load_library_mem(g->ctx[(size_t)r], lib, n, names, seqs, offs); });
}

// One sample's reads from memory: rank r takes the contiguous chunk [n r / W, n (r + 1) / W) ("contiguous read chunks")
int32_t mirge_group_collapse_seqs(mirge_group* g, const char* seqs, const uint64_t* offs, uint64_t n, int32_t sample) {
    if (!g || !offs || (n && !seqs)) return g ? gfail(g, MIRGE_E_ARG, "null input") : MIRGE_E_ARG;
    g->merged = false; g->annotated = false;
    const uint64_t W = (uint64_t)g->world;
    return group_parallel(g, [&](int r) {
        const uint64_t lo = n * (uint64_t)r / W, hi = n * ((uint64_t)r + 1) / W;
        return (int)mirge_collapse_seqs(g->ctx[(size_t)r], seqs, offs + lo, hi - lo, sample);
    });
}

// Many sample files at once: file i goes to rank i mod W as a whole ("whole samples per GPU"), every rank
// working through its files in order with its share of the host's threads reading / inflating ahead.
int32_t mirge_group_collapse_fastq_batch(mirge_group* g, int32_t n, const char* const* paths, const int32_t* samples, const char* adapter,
                                         int32_t trim5, int32_t trim3, uint64_t* total_reads, uint64_t* trimmed_reads, int32_t* phred) {
    if (!g || n < 0 || (n && (!paths || !samples))) return g ? gfail(g, MIRGE_E_ARG, "null input") : MIRGE_E_ARG;
    for (int i = 0; i < n; ++i) if (samples[i] < 0 || samples[i] >= g->S || !paths[i]) return gfail(g, MIRGE_E_ARG, "sample index out of range or null path");
    g->merged = false; g->annotated = false;
    const int W = g->world, per = std::max(1, host_threads() / W);
    return group_parallel(g, [&](int r) {
        std::vector<int> idx;
        for (int i = r; i < n; i += W) idx.push_back(i);
        return collapse_files(g->ctx[(size_t)r], idx, paths, samples, adapter, trim5, trim3, total_reads, trimmed_reads, phred, per);
    });
}

int32_t mirge_group_annotate(mirge_group* g) {
    if (!g) return MIRGE_E_ARG;
    int rc = MIRGE_OK;
    if (g->world > 1 || !g->staged) rc = group_merge(g);          // (one rank through RCCL still makes the calls: copies to itself)
    if (rc) return rc;
    rc = group_parallel(g, [&](int r) { return (int)mirge_annotate(g->ctx[(size_t)r]); });
    if (rc) return rc;
    rc = group_allreduce(g);
    if (!rc) g->annotated = true;
    return rc;
}

int32_t mirge_group_counts(mirge_group* g, const uint64_t** quant, const uint64_t** iscan, const uint64_t** category,
                           const uint64_t** len_hist, const uint64_t** trimmed_uniq) {
    if (!g) return MIRGE_E_ARG;
    if (!g->annotated) return gfail(g, MIRGE_E_ARG, "mirge_group_counts before mirge_group_annotate");
    const int rc = mirge_counts(g->ctx[0], quant, iscan, category, len_hist, trimmed_uniq);      // (every rank holds the sums)
    return rc ? gfail(g, rc, g->ctx[0]->err) : MIRGE_OK;
}
int32_t mirge_group_round_stats(const mirge_group* g, int32_t round, uint64_t* processed, uint64_t* aligned, double* seconds) {
    if (!g || !g->annotated) return MIRGE_E_ARG;
    int rc = mirge_round_stats(g->ctx[0], round, processed, aligned, nullptr);
    if (rc) return rc;
    if (seconds) { double mx = 0; for (auto* c : g->ctx) { double s = 0; (void)mirge_round_stats(c, round, nullptr, nullptr, &s); mx = std::max(mx, s); } *seconds = mx; }
    return MIRGE_OK;
}
// The four tables of the whole job: miR.Counts / miR.RPM from the reduced sums (every rank holds them), mapped / unmapped
// = every rank's rows, one after the other (after the merge a sequence lives on exactly one rank; their order in the
// reference is Perl hash order, miRge.pl:714, 851).
int32_t mirge_group_write_tables(mirge_group* g, const char* merges_csv, const char* output_dir, const char* const* sample_names,
                                 uint64_t* mirna_reads_filtered, uint64_t* mirna_uniq_filtered) {
    if (!g || !output_dir || !sample_names) return g ? gfail(g, MIRGE_E_ARG, "null argument") : MIRGE_E_ARG;
    if (!g->annotated) return gfail(g, MIRGE_E_ARG, "mirge_group_annotate has not run");
    const std::string dir = output_dir;
    (void)mkdir(dir.c_str(), 0777);
    std::vector<std::string> sub((size_t)g->world);
    for (int r = 0; r < g->world; ++r) sub[(size_t)r] = r == 0 ? dir : dir + "/.rank" + std::to_string(r);
    int rc = group_parallel(g, [&](int r) {
        return (int)mirge_write_tables(g->ctx[(size_t)r], merges_csv, sub[(size_t)r].c_str(), sample_names,
                                       r == 0 ? mirna_reads_filtered : nullptr, r == 0 ? mirna_uniq_filtered : nullptr);
    });
    for (int r = 1; r < g->world; ++r) {
        for (const char* nm : {"mapped.csv", "unmapped.csv"}) {
            std::string data;
            if (!rc) {
                FILE* f = fopen((sub[(size_t)r] + "/" + nm).c_str(), "rb");
                FILE* o = fopen((dir + "/" + nm).c_str(), "ab");
                if (!f || !o) rc = gfail(g, MIRGE_E_IO, "cannot merge the per-rank tables in " + dir);
                else {
                    int ch; while ((ch = fgetc(f)) != EOF && ch != '\n') {}          // the header line is rank 0's
                    char buf[1 << 16]; size_t k;
                    while ((k = fread(buf, 1, sizeof buf, f)) > 0) if (fwrite(buf, 1, k, o) != k) { rc = gfail(g, MIRGE_E_IO, "short write in " + dir); break; }
                }
                if (f) fclose(f);
                if (o && fclose(o) != 0 && !rc) rc = gfail(g, MIRGE_E_IO, "short write in " + dir);
            }
        }
        for (const char* nm : {"mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv"}) (void)unlink((sub[(size_t)r] + "/" + nm).c_str());
        (void)rmdir(sub[(size_t)r].c_str());
    }
    return rc;
}

uint64_t mirge_group_n_unique(const mirge_group* g) { uint64_t n = 0; if (g) for (auto* c : g->ctx) n += mirge_n_unique(c); return n; }
int32_t mirge_group_timing(const mirge_group* g, double* merge_ms, double* allreduce_ms) {
    if (!g) return MIRGE_E_ARG;
    if (merge_ms) *merge_ms = g->merge_ms;
    if (allreduce_ms) *allreduce_ms = g->allreduce_ms;
    return MIRGE_OK;
}

// Many sample files through ONE context, in order, with the host's threads reading / inflating files ahead of the device
// (a gzip stream inflates serially; different files do not wait for each other).  adapter = NULL: trimmed files, as
// mirge_collapse_fastq; else raw files through the trimming step, as mirge_collapse_raw_fastq.
int32_t mirge_collapse_fastq_batch(mirge_ctx* c, int32_t n, const char* const* paths, const int32_t* samples, const char* adapter,
                                   int32_t trim5, int32_t trim3, uint64_t* total_reads, uint64_t* trimmed_reads, int32_t* phred) {
    if (!c || n < 0 || (n && (!paths || !samples))) return c ? fail(c, MIRGE_E_ARG, "null input") : MIRGE_E_ARG;
    for (int i = 0; i < n; ++i) if (samples[i] < 0 || samples[i] >= c->S || !paths[i]) return fail(c, MIRGE_E_ARG, "sample index out of range or null path");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int> idx((size_t)n);
    for (int i = 0; i < n; ++i) idx[(size_t)i] = i;
    return collapse_files(c, idx, paths, samples, adapter, trim5, trim3, total_reads, trimmed_reads, phred, host_threads());
}

}  // extern "C"
