// orbhip_pool.hip — one node, G GPUs (BASELINE.json configs 4 and 5): the product-side owner of "one context per GPU,
// one host thread per GPU, pinned staging ring per GPU, no collective".
//
// The reference's only concurrency on this path is the pair of std::threads that run the left and right extractor of a stereo Frame
// (Frame.cc:78-81).  A pool is that idea at node scale: devices[r] gets a worker thread that owns an extractor context (camera c is
// served by devices[c mod G]) and, for the relocalisation query of config 5, the rows [lo_r, hi_r) of the descriptor DB.  Workers never
// exchange data: a round of frames is G independent submit/collect pairs on the pipelined host path (orbhip_api.hip), a DB query is G
// independent uploads of the same 64 KB query + G shard scans whose per-query (best, second, index) triples are merged on the host
// with the matcher's rule — 2000 x G x 16 B, not worth an all-gather over xGMI.
#include "orbhip_internal.h"
#include <sched.h>
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Worker {
    int device = 0, index = 0, numa_node = -1; bool numa_bound = false;
    orbhip_ctx* ctx = nullptr;
    std::thread th; std::mutex m; std::condition_variable cv, cv_done; std::deque<std::function<void()>> q; int pending = 0; bool stop = false;
    // result of the last task(s)
    orbhip_status st = ORBHIP_OK; std::string err;
    // frames: per pool ticket (mod ring) the context's ticket and the cameras it covered
    int ctx_ticket[8]; std::vector<int> cams[8]; orbhip_status slot_st[8] = {}; std::string slot_err[8];
    // descriptor DB shard + query staging
    uint8_t* d_db = nullptr; int64_t lo = 0, hi = 0;
    uint8_t* d_dbx = nullptr;          // the shard expanded once for the FP4 scan (orbhip_nn_expand_device): 128 B per row beside the 32; nullptr = scan the bit form
    hipStream_t qstream = nullptr; uint8_t* d_q = nullptr; long long* d_bi = nullptr; int* d_bd = nullptr; int* d_sd = nullptr; int q_cap = 0;
    uint8_t* h_q = nullptr; long long* h_bi = nullptr; int* h_bd = nullptr; int* h_sd = nullptr;

    void post(std::function<void()> f) { { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(f)); pending++; } cv.notify_one(); }
    void wait() { std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [&] { return pending == 0; }); }
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); }
            f();
            { std::lock_guard<std::mutex> lk(m); pending--; }
            cv_done.notify_all();
        }
    }
    void fail_from_thread(orbhip_status s) { st = s; err = orbhip_last_error(); }
    void fail_hip(const char* what, hipError_t e) { st = ORBHIP_ERR_HIP; err = std::string(what) + ": " + hipGetErrorString(e); }
};

}  // namespace

struct orbhip_pool {
    std::vector<std::unique_ptr<Worker>> w; int ncam = 0, per_dev = 0, cap = 0; orbhip_config cfg;
    int next_ticket = 0, oldest_ticket = 0; int64_t ndb = 0;
};

static orbhip_status pool_status(orbhip_pool* p, const char* what)
{   // first failure of the round, reported on the calling thread; EVERY worker's status is cleared with it (a round that all devices refuse must
    // not leave the second device's refusal behind to be blamed on the next, healthy round)
    orbhip_status first = ORBHIP_OK; int dev = -1; std::string msg;
    for (auto& w : p->w) {
        if (w->st != ORBHIP_OK && first == ORBHIP_OK) { first = w->st; dev = w->device; msg = w->err; }
        w->st = ORBHIP_OK; w->err.clear();
    }
    return first == ORBHIP_OK ? ORBHIP_OK : orbhip_set_error(first, "%s (device %d): %s", what, dev, msg.c_str());
}

// ---- NUMA placement.  On a two-socket node every GPU hangs off one socket's PCIe root: a worker whose copy loops (pageable -> pinned gathers,
// result scatters) and pinned ring live on the other socket pay a cross-socket hop on every byte, and at 8 GPUs x ~55 GB/s of uploads the host side
// is what bends the scaling curve.  Each worker therefore binds itself to the CPUs of its device's NUMA node BEFORE it creates its context:
// the context's pinned mirrors (hipHostMalloc, first touched by this thread) then come from that node's memory.  ORBHIP_POOL_NUMA=0 turns it off.
// (orbhip_device_numa_node / orbhip_bind_thread_to_node live in orbhip_api.hip: the host path's copy helpers are placed with them too)

extern "C" void orbhip_pool_destroy(orbhip_pool* p)
{
    if (!p) return;
    for (auto& w : p->w) {
        if (!w->th.joinable()) continue;
        w->post([wp = w.get()] {
            (void)hipSetDevice(wp->device);
            if (wp->ctx) orbhip_destroy(wp->ctx);
            if (wp->qstream) { (void)hipStreamSynchronize(wp->qstream); (void)hipStreamDestroy(wp->qstream); }
            void* d[] = {wp->d_db, wp->d_dbx, wp->d_q, wp->d_bi, wp->d_bd, wp->d_sd}; for (void* x : d) if (x) (void)hipFree(x);
            void* h[] = {wp->h_q, wp->h_bi, wp->h_bd, wp->h_sd}; for (void* x : h) if (x) (void)hipHostFree(x);
        });
        w->wait();
        { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
        w->cv.notify_all();
        w->th.join();
    }
    delete p;
}

extern "C" orbhip_status orbhip_pool_create(orbhip_pool** out, const int* devices, int ndevices, const orbhip_config* cfg, int ncameras)
{
    if (!out || !devices || !cfg || ndevices < 1 || ndevices > 64 || ncameras < 1) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return orbhip_set_error(ORBHIP_ERR_HIP, "no HIP device available: the ORB front-end has no CPU fallback");
    for (int r = 0; r < ndevices; r++) if (devices[r] < 0 || devices[r] >= ndev) return orbhip_set_error(ORBHIP_ERR_INVALID, "device %d out of range: %d HIP device(s) visible", devices[r], ndev);
    orbhip_pool* p = new orbhip_pool; p->cfg = *cfg; p->ncam = ncameras; p->per_dev = (ncameras + ndevices - 1) / ndevices;
    for (int r = 0; r < ndevices; r++) {
        std::unique_ptr<Worker> w(new Worker); w->device = devices[r]; w->index = r;
        for (int& t : w->ctx_ticket) t = -1;
        Worker* wp = w.get();
        w->th = std::thread([wp] { wp->loop(); });
        orbhip_config c = *cfg; c.device = devices[r]; c.max_batch = p->per_dev; c.stream = nullptr;
        w->post([wp, c] {
            static const bool numa = [] { const char* e = getenv("ORBHIP_POOL_NUMA"); return !(e && *e == '0'); }();
            if (numa) { wp->numa_node = orbhip_device_numa_node(wp->device); wp->numa_bound = orbhip_bind_thread_to_node(wp->numa_node); }
            const orbhip_status s = orbhip_create(&wp->ctx, &c); if (s != ORBHIP_OK) wp->fail_from_thread(s);
        });
        p->w.push_back(std::move(w));
    }
    for (auto& w : p->w) w->wait();
    const orbhip_status st = pool_status(p, "orbhip_pool_create");
    if (st != ORBHIP_OK) { orbhip_pool_destroy(p); return st; }
    p->cap = orbhip_keypoint_capacity(p->w[0]->ctx);
    *out = p;
    return ORBHIP_OK;
}

extern "C" int orbhip_pool_num_devices(const orbhip_pool* p) { return p ? (int)p->w.size() : 0; }
extern "C" int orbhip_pool_device_of(const orbhip_pool* p, int camera) { return (p && camera >= 0 && camera < p->ncam) ? p->w[camera % p->w.size()]->device : -1; }
extern "C" int orbhip_pool_keypoint_capacity(const orbhip_pool* p) { return p ? p->cap : 0; }
extern "C" int orbhip_pool_numa_node(const orbhip_pool* p, int r, int* bound)
{
    if (!p || r < 0 || r >= (int)p->w.size()) { if (bound) *bound = 0; return -1; }
    if (bound) *bound = p->w[r]->numa_bound ? 1 : 0;
    return p->w[r]->numa_node;
}

extern "C" orbhip_status orbhip_pool_submit(orbhip_pool* p, const uint8_t* const* imgs, int stride, int* ticket)
{
    if (!p || !imgs || !ticket) return orbhip_set_error(ORBHIP_ERR_INVALID, "null argument");
    if (p->next_ticket - p->oldest_ticket >= orbhip_ring_depth()) return orbhip_set_error(ORBHIP_ERR_INVALID, "ring full: %d rounds in flight, collect ticket %d first", orbhip_ring_depth(), p->oldest_ticket);
    const int G = (int)p->w.size(), slot = p->next_ticket % 8;
    for (int r = 0; r < G; r++) {
        Worker* w = p->w[r].get();
        w->cams[slot].clear(); w->ctx_ticket[slot] = -1;
        for (int c = r; c < p->ncam; c += G) if (imgs[c]) w->cams[slot].push_back(c);           // camera c -> devices[c mod G]
        if (w->cams[slot].empty()) continue;
        w->post([w, slot, imgs, stride] {
            std::vector<const uint8_t*> mine; for (int c : w->cams[slot]) mine.push_back(imgs[c]);
#ifdef ORBHIP_TEST_HOOKS      // the CPU emulation build only (tests/emu): ORBHIP_TEST_FAIL_SUBMIT_WORKER=<r> makes worker r refuse its submits (partial-failure path)
            const char* inj = getenv("ORBHIP_TEST_FAIL_SUBMIT_WORKER");
            if (inj && *inj && atoi(inj) == w->index) { w->ctx_ticket[slot] = -1; w->st = ORBHIP_ERR_HIP; w->err = "injected submit failure"; return; }
#endif
            const orbhip_status s = orbhip_submit(w->ctx, (int)mine.size(), mine.data(), stride, &w->ctx_ticket[slot]);
            if (s != ORBHIP_OK) { w->ctx_ticket[slot] = -1; w->fail_from_thread(s); }
        });
    }
    for (auto& w : p->w) w->wait();                       // pageable frames are staged (in parallel, one thread per device) when this returns
    // A device that refused its part of the round: tickets are collected strictly in submission order, so the parts the other devices did
    // take cannot be withdrawn (older rounds may still be in flight before them).  The round therefore gets its pool ticket all the same —
    // every context ticket has an owner — and the refusal is reported by orbhip_pool_collect of that ticket (the refused cameras deliver
    // nothing).  Only a round that NO device took fails here, without a ticket.
    bool any_taken = false, any_failed = false;
    for (auto& w : p->w) { any_taken = any_taken || w->ctx_ticket[slot] >= 0; any_failed = any_failed || w->st != ORBHIP_OK; }
    if (any_failed && !any_taken) return pool_status(p, "orbhip_pool_submit");
    for (auto& w : p->w) { w->slot_st[slot] = w->st; w->slot_err[slot] = w->err; w->st = ORBHIP_OK; w->err.clear(); }
    *ticket = p->next_ticket++;
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_pool_collect(orbhip_pool* p, int ticket, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!p || !n_out || cap < 0) return orbhip_set_error(ORBHIP_ERR_INVALID, "null argument");
    if (ticket != p->oldest_ticket || ticket >= p->next_ticket) return orbhip_set_error(ORBHIP_ERR_INVALID, "ticket %d is not the oldest round in flight (%d)", ticket, p->oldest_ticket);
    const int slot = ticket % 8;
    for (int c = 0; c < p->ncam; c++) n_out[c] = 0;
    for (auto& wu : p->w) {
        Worker* w = wu.get();
        if (w->ctx_ticket[slot] < 0) continue;
        w->post([w, slot, kps, desc, cap, n_out] {
            const std::vector<int>& cams = w->cams[slot];
            std::vector<orbhip_keypoint*> pk; std::vector<uint8_t*> pd; std::vector<int*> pn;
            for (int c : cams) { pk.push_back(kps ? kps + (size_t)c * cap : nullptr); pd.push_back(desc ? desc + (size_t)c * cap * 32 : nullptr); pn.push_back(n_out + c); }
            const orbhip_status s = orbhip_collect_scatter(w->ctx, w->ctx_ticket[slot], pk.data(), pd.data(), cap, pn.data());
            if (s != ORBHIP_OK) w->fail_from_thread(s);
        });
    }
    for (auto& w : p->w) w->wait();
    p->oldest_ticket++;
    const orbhip_status st = pool_status(p, "orbhip_pool_collect"); if (st != ORBHIP_OK) return st;
    for (auto& w : p->w)                                  // a device that had refused this round at submit time
        if (w->slot_st[slot] != ORBHIP_OK) { const orbhip_status s = w->slot_st[slot]; w->slot_st[slot] = ORBHIP_OK; return orbhip_set_error(s, "orbhip_pool_submit of this round (device %d): %s", w->device, w->slot_err[slot].c_str()); }
    return ORBHIP_OK;
}

extern "C" orbhip_status orbhip_pool_extract(orbhip_pool* p, const uint8_t* const* imgs, int stride, orbhip_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    int t = -1;
    const orbhip_status st = orbhip_pool_submit(p, imgs, stride, &t); if (st != ORBHIP_OK) return st;
    return orbhip_pool_collect(p, t, kps, desc, cap, n_out);
}

// ---------------------------------------------------------------------------------------------- descriptor DB shards (config 5)
extern "C" void orbhip_pool_db_shard(const orbhip_pool* p, int r, int64_t* lo, int64_t* hi)
{
    int64_t a = 0, b = 0;
    if (p && r >= 0 && r < (int)p->w.size()) {
        const int64_t G = (int64_t)p->w.size(), base = p->ndb / G, rem = p->ndb % G;          // contiguous ranges whose sizes differ by at most one row
        a = r * base + std::min<int64_t>(r, rem); b = a + base + (r < rem ? 1 : 0);
    }
    if (lo) *lo = a; if (hi) *hi = b;
}

extern "C" orbhip_status orbhip_pool_db_load(orbhip_pool* p, const uint8_t* db, int64_t ndb)
{
    if (!p || ndb < 0 || (ndb > 0 && !db)) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    p->ndb = ndb;
    for (int r = 0; r < (int)p->w.size(); r++) {
        Worker* w = p->w[r].get();
        orbhip_pool_db_shard(p, r, &w->lo, &w->hi);
        w->post([w, db] {
            hipError_t e = hipSetDevice(w->device);
            if (e == hipSuccess && w->d_db) { e = hipFree(w->d_db); w->d_db = nullptr; }
            if (e == hipSuccess && w->d_dbx) { e = hipFree(w->d_dbx); w->d_dbx = nullptr; }
            const size_t bytes = (size_t)(w->hi - w->lo) * 32;
            if (e == hipSuccess) e = orbhip_dmalloc((void**)&w->d_db, std::max<size_t>(bytes, 32));
            if (e == hipSuccess && bytes) e = hipMemcpy(w->d_db, db + (size_t)w->lo * 32, bytes, hipMemcpyHostToDevice);
            if (e == hipSuccess && !w->qstream) e = hipStreamCreateWithFlags(&w->qstream, hipStreamNonBlocking);
            // A database is loaded once and asked many times: from the size on at which a query takes the matrix-core scan, the shard is also kept in the form
            // that scan multiplies (four times the bytes; ORBHIP_POOL_DB_EXPAND=0 or no memory for it: the bit form is scanned, the answers are the same)
            const char* ex = getenv("ORBHIP_POOL_DB_EXPAND");
            if (e == hipSuccess && w->hi - w->lo >= 4 * 8192 && !(ex && ex[0] == '0')) {
                if (orbhip_dmalloc((void**)&w->d_dbx, orbhip_nn_expanded_size(w->hi - w->lo)) != hipSuccess) { w->d_dbx = nullptr; (void)hipGetLastError(); }
                else if (orbhip_nn_expand_device(w->qstream, w->d_db, w->hi - w->lo, w->d_dbx) != ORBHIP_OK || hipStreamSynchronize(w->qstream) != hipSuccess) { (void)hipFree(w->d_dbx); w->d_dbx = nullptr; }
            }
            if (e != hipSuccess) w->fail_hip("descriptor DB shard upload", e);
        });
    }
    for (auto& w : p->w) w->wait();
    return pool_status(p, "orbhip_pool_db_load");
}

extern "C" orbhip_status orbhip_pool_db_query(orbhip_pool* p, const uint8_t* q, int nq, int64_t* best_idx, int32_t* best_dist, int32_t* second_dist)
{
    if (!p || nq < 0 || (nq > 0 && (!q || !best_idx || !best_dist || !second_dist))) return orbhip_set_error(ORBHIP_ERR_INVALID, "bad argument");
    if (nq == 0) return ORBHIP_OK;
    for (auto& wu : p->w) {
        Worker* w = wu.get();
        if (!w->d_db) return orbhip_set_error(ORBHIP_ERR_INVALID, "no descriptor DB: call orbhip_pool_db_load first");
        w->post([w, q, nq] {
            hipError_t e = hipSetDevice(w->device);
            if (e == hipSuccess && w->q_cap < nq) {
                void* d[] = {w->d_q, w->d_bi, w->d_bd, w->d_sd}; for (void* x : d) if (x) (void)hipFree(x);
                void* h[] = {w->h_q, w->h_bi, w->h_bd, w->h_sd}; for (void* x : h) if (x) (void)hipHostFree(x);
                w->d_q = nullptr; w->d_bi = nullptr; w->d_bd = nullptr; w->d_sd = nullptr; w->h_q = nullptr; w->h_bi = nullptr; w->h_bd = nullptr; w->h_sd = nullptr; w->q_cap = 0;
                const size_t n = (size_t)nq + nq / 4;
                if (e == hipSuccess) e = orbhip_dmalloc((void**)&w->d_q, n * 32);
                if (e == hipSuccess) e = orbhip_dmalloc((void**)&w->d_bi, n * 8);
                if (e == hipSuccess) e = orbhip_dmalloc((void**)&w->d_bd, n * 4);
                if (e == hipSuccess) e = orbhip_dmalloc((void**)&w->d_sd, n * 4);
                if (e == hipSuccess) e = hipHostMalloc((void**)&w->h_q, n * 32, hipHostMallocDefault);
                if (e == hipSuccess) e = hipHostMalloc((void**)&w->h_bi, n * 8, hipHostMallocDefault);
                if (e == hipSuccess) e = hipHostMalloc((void**)&w->h_bd, n * 4, hipHostMallocDefault);
                if (e == hipSuccess) e = hipHostMalloc((void**)&w->h_sd, n * 4, hipHostMallocDefault);
                if (e == hipSuccess) w->q_cap = (int)n;
            }
            if (e != hipSuccess) { w->fail_hip("descriptor DB query buffers", e); return; }
            memcpy(w->h_q, q, (size_t)nq * 32);                                                    // the "broadcast": G independent 64 KB copies
            e = hipMemcpyAsync(w->d_q, w->h_q, (size_t)nq * 32, hipMemcpyHostToDevice, w->qstream);
            if (e == hipSuccess && (w->d_dbx ? orbhip_hamming_nn_device_expanded(w->qstream, w->d_q, nq, w->d_db, w->d_dbx, w->hi - w->lo, w->lo, (int64_t*)w->d_bi, w->d_bd, w->d_sd)
                                             : orbhip_hamming_nn_device(w->qstream, w->d_q, nq, w->d_db, w->hi - w->lo, w->lo, (int64_t*)w->d_bi, w->d_bd, w->d_sd)) != ORBHIP_OK) { w->fail_from_thread(ORBHIP_ERR_HIP); return; }
            if (e == hipSuccess) e = hipMemcpyAsync(w->h_bi, w->d_bi, (size_t)nq * 8, hipMemcpyDeviceToHost, w->qstream);
            if (e == hipSuccess) e = hipMemcpyAsync(w->h_bd, w->d_bd, (size_t)nq * 4, hipMemcpyDeviceToHost, w->qstream);
            if (e == hipSuccess) e = hipMemcpyAsync(w->h_sd, w->d_sd, (size_t)nq * 4, hipMemcpyDeviceToHost, w->qstream);
            if (e == hipSuccess) e = hipStreamSynchronize(w->qstream);
            if (e != hipSuccess) w->fail_hip("descriptor DB query", e);
        });
    }
    for (auto& w : p->w) w->wait();
    const orbhip_status st = pool_status(p, "orbhip_pool_db_query"); if (st != ORBHIP_OK) return st;
    // merge in ascending shard (= ascending global index) order: `if (d < best) { second = best; best = d; idx = i; } else if (d < second) second = d;`
    // of one left-to-right scan (ORBmatcher.cc:447-456 idiom): an equal distance keeps the earlier shard's index
    for (int i = 0; i < nq; i++) {
        long long idx = p->w[0]->h_bi[i]; int best = p->w[0]->h_bd[i], second = p->w[0]->h_sd[i];
        for (size_t r = 1; r < p->w.size(); r++) {
            const Worker& w = *p->w[r];
            if (w.hi == w.lo) continue;
            const int pb = w.h_bd[i], ps = w.h_sd[i];
            if (pb < best) { second = std::min(best, ps); best = pb; idx = w.h_bi[i]; }
            else second = std::min(second, pb);
        }
        best_idx[i] = idx; best_dist[i] = best; second_dist[i] = second;
    }
    return ORBHIP_OK;
}
