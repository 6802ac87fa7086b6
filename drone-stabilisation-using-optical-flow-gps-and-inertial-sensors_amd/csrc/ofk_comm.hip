// ofk_comm.hip — the one exchange step of the multi-GPU path: RCCL over xGMI, bound at run time (dlopen of librccl.so, so a
// single-GPU user needs no RCCL, and the build needs no RCCL headers) and driven from the library's own streams.  One process per GPU; ranks own independent frame
// pairs (no data-path collective); after every step the [B, 8] f32 velocity records of all ranks are all-gathered, stream-ordered
// behind the step's solve on the last slice's stream — no host wait, no second HIP runtime in the process.
#include "ofk_internal.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// The few RCCL declarations this file needs, stated here so that the build does not depend on the RCCL headers (the library is
// bound with dlopen at run time; a single-GPU user needs neither).  Values and layouts are the stable NCCL 2 ABI that
// rccl/rccl.h declares: ncclUniqueId = 128 opaque bytes passed BY VALUE to ncclCommInitRank, ncclSuccess = 0,
// ncclFloat32 = 7, ncclFloat64 = 8, ncclSum / ncclMax / ncclMin = 0 / 2 / 3.
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum { ncclSuccess = 0, ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0, ncclMax = 2, ncclMin = 3 };

struct ofk_comm {
    void *lib;
    ncclResult_t (*get_unique_id)(ncclUniqueId *);
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*comm_destroy)(ncclComm_t);
    ncclResult_t (*all_gather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char *(*error_string)(ncclResult_t);
    ncclComm_t comm[OFK_MAX_STREAMS]; int ncomm;   // one communicator per free-running slice: a slice gathers its own records on its
                                                    // own stream, so the exchange never couples the slices (comm[0] also serves all-reduces)
    int rank, world;
    float *send[2], *recv[2];           // [B][8] and [world][B][8] f32, two slots so step k+1 may export while step k travels
    int cap_batch;
    hipEvent_t done[2][OFK_MAX_STREAMS];            // gather of slot s (slice k) complete
    hipStream_t xs; hipEvent_t ev_ready;            // one slice: export + all-gather run on a stream of their OWN behind the step's solve, so
                                                    // that the next step's first kernel does not queue behind a collective (xs = exchange stream)
    int slot_slices[2], slot_batch[2];              // how the slot's latest gather was cut
    float *hrecv; size_t hrecv_bytes;               // pinned host staging for ofk_comm_fetch_records
    double *red; void *hred;            // all-reduce scratch (device / pinned host), 64 doubles
};

static ofk_comm g_lib_only;             // symbols for ofk_comm_unique_id (no context needed)

static int load_rccl(ofk_ctx *c, ofk_comm *m)
{
    if (m->lib) return OFK_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) { m->lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (m->lib) break; }
    if (!m->lib) return ofk_fail(c, OFK_E_INVALID, "RCCL not found (dlopen librccl.so): %s", dlerror());
#define SYM(field, name) do { *(void **)(&m->field) = dlsym(m->lib, name); if (!m->field) return ofk_fail(c, OFK_E_INVALID, "librccl.so lacks %s", name); } while (0)
    SYM(get_unique_id, "ncclGetUniqueId"); SYM(comm_init_rank, "ncclCommInitRank"); SYM(comm_destroy, "ncclCommDestroy");
    SYM(all_gather, "ncclAllGather"); SYM(all_reduce, "ncclAllReduce"); SYM(error_string, "ncclGetErrorString");
#undef SYM
    return OFK_OK;
}

#define OFK_NCCL(c, m, call)                                                                                  \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess) return ofk_fail(c, OFK_E_HIP, "%s: %s", #call, (m)->error_string(r_));         \
    } while (0)

extern "C" int ofk_comm_unique_id(uint8_t *ids, int n_ids)
{
    if (!ids || n_ids < 1 || n_ids > OFK_MAX_STREAMS) return ofk_fail(nullptr, OFK_E_INVALID, "ofk_comm_unique_id: 1..%d ids", OFK_MAX_STREAMS);
    int rc = load_rccl(nullptr, &g_lib_only);
    if (rc != OFK_OK) return rc;
    for (int k = 0; k < n_ids; ++k) {
        ncclUniqueId id;
        OFK_NCCL(nullptr, &g_lib_only, g_lib_only.get_unique_id(&id));
        memcpy(ids + (size_t)k * NCCL_UNIQUE_ID_BYTES, id.internal, NCCL_UNIQUE_ID_BYTES);
    }
    return OFK_OK;
}

extern "C" int ofk_comm_init(ofk_ctx *c, const uint8_t *ids, int n_ids, int rank, int world)
{
    if (!c || !ids || n_ids < 1 || n_ids > OFK_MAX_STREAMS || world < 1 || rank < 0 || rank >= world) return ofk_fail(c, OFK_E_INVALID, "ofk_comm_init: bad argument");
    if (c->comm) return ofk_fail(c, OFK_E_INVALID, "ofk_comm_init: the context already has a communicator");
    OFK_HIP(c, hipSetDevice(c->device));
    ofk_comm *m = (ofk_comm *)calloc(1, sizeof(ofk_comm));
    if (!m) return ofk_fail(c, OFK_E_INVALID, "out of host memory");
    int rc = load_rccl(c, m);
    if (rc != OFK_OK) { free(m); return rc; }
    c->comm = m;
    // the pipeline's own streams first: RCCL creates streams at ncclCommInitRank, and the runtime deals hardware queues in
    // creation order (see ofk_set_streams)
    rc = ofk_prepare_streams(c);
    if (rc != OFK_OK) { ofk_comm_destroy(c); return rc; }
    if (hipStreamCreateWithFlags(&m->xs, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->ev_ready, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        ofk_comm_destroy(c);
        return ofk_fail(c, OFK_E_HIP, "ofk_comm_init: exchange stream");
    }
    {   // communicator 0: the one every rank needs (gathers with one slice, all-reduces, barriers)
        ncclUniqueId id;
        memcpy(id.internal, ids, NCCL_UNIQUE_ID_BYTES);
        ncclResult_t r = m->comm_init_rank(&m->comm[0], world, id, rank);
        if (r != ncclSuccess) {
            ofk_fail(c, OFK_E_HIP, "ncclCommInitRank(communicator 0, rank %d of %d): %s", rank, world, m->error_string(r));
            ofk_comm_destroy(c);
            return OFK_E_HIP;
        }
        m->ncomm = 1;
    }
    m->rank = rank; m->world = world; m->cap_batch = c->max_batch;
    const size_t sb = (size_t)c->max_batch * 8 * sizeof(float);
    bool ok = true;
    for (int s = 0; s < 2 && ok; ++s) {
        ok = hipMalloc((void **)&m->send[s], sb) == hipSuccess && hipMalloc((void **)&m->recv[s], sb * world) == hipSuccess;
        for (int k = 0; k < OFK_MAX_STREAMS && ok; ++k) ok = hipEventCreateWithFlags(&m->done[s][k], hipEventDisableTiming) == hipSuccess;
    }
    m->hrecv_bytes = sb * world;
    ok = ok && hipMalloc((void **)&m->red, 64 * sizeof(double)) == hipSuccess && hipHostMalloc(&m->hred, 64 * sizeof(double)) == hipSuccess &&
         hipHostMalloc((void **)&m->hrecv, m->hrecv_bytes) == hipSuccess;
    if (!ok) { ofk_comm_destroy(c); return ofk_fail(c, OFK_E_HIP, "ofk_comm_init: device buffers"); }
    // Per-slice communicators (n_ids > 1).  ncclCommInitRank is collective: a rank that gave up on communicator k while its peers were
    // inside that call would leave them waiting for ever, so the COUNT is agreed first - the smallest n_ids any rank passed, one
    // all-reduce over communicator 0 - then every rank creates exactly that many, and a failure from here on is fatal (error return,
    // everything destroyed; the launcher ends the job), never a silent per-rank fallback.
    if (n_ids > 1) {
        double want = (double)n_ids;
        rc = ofk_comm_allreduce_f64(c, &want, 1, 2);
        if (rc != OFK_OK) { ofk_comm_destroy(c); return rc; }
        const int agreed = (int)want < 1 ? 1 : (int)want;
        for (int k = 1; k < agreed; ++k) {
            rc = ofk_comm_add(c, ids + (size_t)k * NCCL_UNIQUE_ID_BYTES);
            if (rc != OFK_OK) { ofk_comm_destroy(c); return rc; }
        }
    }
    return OFK_OK;
}

// One more communicator (a slice of its own to gather on).  Collective: every rank of the world calls it the same number of times with
// the same ids (sharding.Comm agrees on the count over communicator 0 first); an error here cannot be recovered from locally.
extern "C" int ofk_comm_add(ofk_ctx *c, const uint8_t *id_bytes)
{
    if (!c || !c->comm || !id_bytes) return ofk_fail(c, OFK_E_INVALID, "ofk_comm_add: no communicator 0 yet (ofk_comm_init) or NULL id");
    ofk_comm *m = c->comm;
    if (m->ncomm >= OFK_MAX_STREAMS) return ofk_fail(c, OFK_E_INVALID, "ofk_comm_add: at most %d communicators", OFK_MAX_STREAMS);
    OFK_HIP(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = m->comm_init_rank(&m->comm[m->ncomm], m->world, id, m->rank);
    if (r != ncclSuccess) {
        m->comm[m->ncomm] = nullptr;
        return ofk_fail(c, OFK_E_HIP, "ncclCommInitRank(communicator %d, rank %d of %d): %s - the ranks had agreed on this communicator, so this is fatal "
                        "for the job (run with one communicator: --comms 1)", m->ncomm, m->rank, m->world, m->error_string(r));
    }
    ++m->ncomm;
    return OFK_OK;
}

extern "C" int ofk_comm_destroy(ofk_ctx *c)
{
    if (!c || !c->comm) return OFK_OK;
    ofk_comm *m = c->comm;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (int k = 0; k < m->ncomm; ++k) if (m->comm[k]) m->comm_destroy(m->comm[k]);
    for (int s = 0; s < 2; ++s) {
        if (m->send[s]) hipFree(m->send[s]);
        if (m->recv[s]) hipFree(m->recv[s]);
        for (int k = 0; k < OFK_MAX_STREAMS; ++k) if (m->done[s][k]) hipEventDestroy(m->done[s][k]);
    }
    if (m->xs) hipStreamDestroy(m->xs);
    if (m->ev_ready) hipEventDestroy(m->ev_ready);
    if (m->red) hipFree(m->red);
    if (m->hred) hipHostFree(m->hred);
    if (m->hrecv) hipHostFree(m->hrecv);
    free(m);
    c->comm = nullptr;
    return OFK_OK;
}

extern "C" int ofk_comm_rank(const ofk_ctx *c) { return c && c->comm ? c->comm->rank : 0; }
extern "C" int ofk_comm_world(const ofk_ctx *c) { return c && c->comm ? c->comm->world : 1; }

// The records of the latest ofk_pairs_run as f32 into send[slot], then ncclAllGather into recv[slot].  Nothing waits on the host.
// With S free-running slices and at least S communicators, every slice exports and gathers ITS pairs on its own stream, right
// behind its own solve: the exchange adds no dependency between the slices (one gather behind the last slice coupled them every
// step and cost 5 % of the rate).  recv[slot] then holds, per slice, [world][pairs of the slice][8]; ofk_comm_fetch_records
// restores the rank-major [world][batch][8] order.  Otherwise: one gather on the stream that ends the step.
extern "C" int ofk_comm_gather_records(ofk_ctx *c, int batch, int slot)
{
    if (!c || !c->comm) return ofk_fail(c, OFK_E_INVALID, "ofk_comm_gather_records: no communicator (ofk_comm_init)");
    ofk_comm *m = c->comm;
    if (batch < 1 || batch > c->cur_batch || batch > m->cap_batch || slot < 0 || slot > 1) return ofk_fail(c, OFK_E_INVALID, "ofk_comm_gather_records: bad argument");
    OFK_HIP(c, hipSetDevice(c->device));
    const int S = c->slices_open ? c->open_slices : 1;
    if (S > 1 && m->ncomm >= S && batch == c->cur_batch) {
        for (int k = 0; k < S; ++k) {
            const int b0 = (int)((long long)batch * k / S), nb = (int)((long long)batch * (k + 1) / S) - b0;
            if (nb <= 0) continue;
            hipStream_t st = k == 0 ? c->stream : c->streams[k];
            ofk_launch_records_f32(st, c->records + (size_t)b0 * OFK_RECORD_DOUBLES, m->send[slot] + (size_t)b0 * 8, nb);
            OFK_NCCL(c, m, m->all_gather(m->send[slot] + (size_t)b0 * 8, m->recv[slot] + (size_t)m->world * b0 * 8, (size_t)nb * 8, ncclFloat32, m->comm[k], st));
            OFK_HIP(c, hipEventRecord(m->done[slot][k], st));
            OFK_HIP(c, hipEventRecord(c->ev_end[k], st));        // the slice's chain now ends behind its gather (join_slices)
        }
        m->slot_slices[slot] = S; m->slot_batch[slot] = batch;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ofk_fail(c, OFK_E_HIP, "k_records_f32: %s", hipGetErrorString(e));
        return OFK_OK;
    }
    if (!c->slices_open) {
        // One slice: the exchange leaves the context's stream.  Behind whatever that stream holds now (the step's solve) the exchange
        // stream exports the records and gathers them; the context's stream goes straight on to the next step, whose solve only waits
        // for the EXPORT kernel (ev_x), never for the collective - no rank's response kernel queues behind another rank's progress.
        OFK_HIP(c, hipEventRecord(m->ev_ready, c->stream));
        OFK_HIP(c, hipStreamWaitEvent(m->xs, m->ev_ready, 0));
        ofk_launch_records_f32(m->xs, c->records, m->send[slot], batch);
        OFK_HIP(c, hipEventRecord(c->ev_x, m->xs));
        c->x_pending = 1;
        OFK_NCCL(c, m, m->all_gather(m->send[slot], m->recv[slot], (size_t)batch * 8, ncclFloat32, m->comm[0], m->xs));
        OFK_HIP(c, hipEventRecord(m->done[slot][0], m->xs));
        m->slot_slices[slot] = 1; m->slot_batch[slot] = batch;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return ofk_fail(c, OFK_E_HIP, "k_records_f32: %s", hipGetErrorString(e));
        return OFK_OK;
    }
    hipStream_t s;                                               // slices open, one communicator: behind the last slice, after the others' end events
    int rc = ofk_export_records_stream(c, m->send[slot], batch, &s);
    if (rc != OFK_OK) return rc;
    OFK_NCCL(c, m, m->all_gather(m->send[slot], m->recv[slot], (size_t)batch * 8, ncclFloat32, m->comm[0], s));
    OFK_HIP(c, hipEventRecord(m->done[slot][0], s));
    OFK_HIP(c, hipEventRecord(c->ev_x, s));                      // the next solves wait for the gather as they did for the export
    c->x_pending = 1;
    m->slot_slices[slot] = 1; m->slot_batch[slot] = batch;
    return OFK_OK;
}

// Waits for the gather of `slot` and copies [world][batch][8] f32 (rank-major) to the host (any rank).
extern "C" int ofk_comm_fetch_records(ofk_ctx *c, int slot, int batch, float *host_out)
{
    if (!c || !c->comm || !host_out || slot < 0 || slot > 1 || batch < 1 || batch > c->comm->cap_batch)
        return ofk_fail(c, OFK_E_INVALID, "ofk_comm_fetch_records: bad argument");
    ofk_comm *m = c->comm;
    if (m->slot_batch[slot] != batch) return ofk_fail(c, OFK_E_INVALID, "ofk_comm_fetch_records: slot %d holds a gather of %d pairs, not %d", slot, m->slot_batch[slot], batch);
    OFK_HIP(c, hipSetDevice(c->device));
    const int S = m->slot_slices[slot];
    for (int k = 0; k < S; ++k) OFK_HIP(c, hipEventSynchronize(m->done[slot][k]));
    const size_t bytes = (size_t)m->world * batch * 8 * sizeof(float);
    if (S == 1) { OFK_HIP(c, hipMemcpy(host_out, m->recv[slot], bytes, hipMemcpyDeviceToHost)); return OFK_OK; }
    OFK_HIP(c, hipMemcpy(m->hrecv, m->recv[slot], bytes, hipMemcpyDeviceToHost));
    return ofk_comm_reorder_records(m->hrecv, m->world, batch, S, host_out);
}

// Host-only (no device, no communicator): the receive buffer of a step gathered per slice holds, slice after slice,
// [world][pairs of the slice][8] f32 - slice k owns the pairs [batch k / S, batch (k+1) / S) of every rank; this restores the
// rank-major [world][batch][8] order ofk_comm_fetch_records returns.  Exported so that the indexing can be checked for any
// world / slice count / uneven cut on a machine without GPUs (tests/test_dist_gloo.py).
extern "C" int ofk_comm_reorder_records(const float *recv, int world, int batch, int slices, float *out)
{
    if (!recv || !out || world < 1 || batch < 1 || slices < 1 || slices > OFK_MAX_STREAMS) return ofk_fail(nullptr, OFK_E_INVALID, "ofk_comm_reorder_records: bad argument");
    for (int k = 0; k < slices; ++k) {
        const int b0 = (int)((long long)batch * k / slices), nb = (int)((long long)batch * (k + 1) / slices) - b0;
        for (int r = 0; r < world; ++r)
            memcpy(out + ((size_t)r * batch + b0) * 8, recv + ((size_t)world * b0 + (size_t)r * nb) * 8, (size_t)nb * 8 * sizeof(float));
    }
    return OFK_OK;
}

// Non-blocking: bit k set = the gather of slice k (bit 0: the single gather) of `slot` has not completed yet; 0 = all done or
// nothing queued; < 0 = error.  For watchdogs: names the communicator a hung step is waiting for.
extern "C" int ofk_comm_pending(ofk_ctx *c, int slot)
{
    if (!c || !c->comm || slot < 0 || slot > 1) return OFK_E_INVALID;
    ofk_comm *m = c->comm;
    int mask = 0;
    for (int k = 0; k < m->slot_slices[slot]; ++k) {
        const hipError_t e = hipEventQuery(m->done[slot][k]);
        if (e == hipErrorNotReady) mask |= 1 << k;
        else if (e != hipSuccess) { (void)hipGetLastError(); return OFK_E_HIP; }
    }
    return mask;
}

extern "C" int ofk_comm_count(const ofk_ctx *c) { return c && c->comm ? c->comm->ncomm : 0; }

// In-place all-reduce of n <= 64 doubles over the ranks (op 0 = sum, 1 = max, 2 = min) on the context's stream, synchronous: the
// benchmark's barrier / max-over-ranks and the Monte-Carlo sweep's (sum v, sum v^2, count) reduction.
extern "C" int ofk_comm_allreduce_f64(ofk_ctx *c, double *inout, int n, int op)
{
    if (!c || !c->comm || !inout || n < 1 || n > 64 || op < 0 || op > 2) return ofk_fail(c, OFK_E_INVALID, "ofk_comm_allreduce_f64: bad argument");
    ofk_comm *m = c->comm;
    OFK_HIP(c, hipSetDevice(c->device));
    int rc = ofk_join_slices(c);
    if (rc != OFK_OK) return rc;
    // never two collectives of one communicator in flight: the gathers of both slots (exchange stream / slice streams) first
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < m->slot_slices[s]; ++k) OFK_HIP(c, hipStreamWaitEvent(c->stream, m->done[s][k], 0));
    memcpy(m->hred, inout, (size_t)n * 8);
    OFK_HIP(c, hipMemcpyAsync(m->red, m->hred, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    OFK_NCCL(c, m, m->all_reduce(m->red, m->red, (size_t)n, ncclFloat64, op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin), m->comm[0], c->stream));
    OFK_HIP(c, hipMemcpyAsync(m->hred, m->red, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(inout, m->hred, (size_t)n * 8);
    return OFK_OK;
}
