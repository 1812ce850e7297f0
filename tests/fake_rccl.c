/* TEST-ONLY transport with librccl's symbols: lets `world` PROCESSES THAT SHARE ONE GPU run the shipped exchange
 * (br_amd/csrc/brx_exchange.hip) with world > 1.  RCCL itself cannot put two ranks on one device, and the boxes this
 * repository reaches have one GPU; libbrx resolves librccl through dlopen (BRX_RCCL_PATH first), so a test selects
 * this library instead and every line of the exchange above the ten nccl* entry points runs unchanged.
 *
 * Semantics kept from RCCL, because the exchange relies on them:
 *   - STREAM-ASYNCHRONOUS, like the real thing: a collective (or the outermost ncclGroupEnd) only ENQUEUES work on the
 *     caller's stream and returns before a byte has moved.  What it enqueues: copies of the outgoing buffers into
 *     page-locked staging (stream order: behind the kernels that fill them), a host function (hipLaunchHostFunc) that
 *     moves staging <-> AF_UNIX sockets and holds the stream until the incoming bytes are there, and the copies of the
 *     incoming staging into the receive buffers.  A kernel the exchange launches on ANOTHER stream, or a host read of a
 *     receive buffer without a synchronisation in between, therefore sees stale bytes -- and the tests see a wrong set;
 *     the first form of this file drained the stream and finished every transfer inside the call, so a missing
 *     dependency could not show.  (FAKE_RCCL_SYNC=1 brings that form back: the stream is synchronised on return.)
 *   - ncclSend / ncclRecv between a pair of ranks match in issue order; inside ncclGroupStart/End they are deferred
 *     and executed together at the outermost GroupEnd (sends on a helper thread, receives on the host function's:
 *     paired send/recv cannot deadlock whatever the socket buffers hold);
 *   - send to self pairs with the recv from self of the same group;
 *   - ncclAllGather works in place (sendbuff == recvbuff + rank * count);
 *   - ncclSum on ncclUint8 WRAPS (the exactness trick of brx_exchange_reduce_counts must hold against that).
 * Differences: bytes travel device -> host -> socket -> host -> device; a message whose size differs from what the
 * receiver posted, or a peer that went away, cannot be returned as an error from a call that has already returned:
 * the host function prints the reason and ABORTS the process (RCCL would hang or corrupt), the test fails loudly.
 *
 * Not product code: nothing under br_amd/ refers to it.  Built by __graft_entry__.build() into tests/libfake_rccl.so. */
#define __HIP_PLATFORM_AMD__ 1
#define _GNU_SOURCE
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#define MAX_WORLD 16
#define MAX_OPS 1024

typedef struct fake_comm {
    int world, rank;
    int fd[MAX_WORLD];
    int listen_fd;
    char base[100];
    uint64_t n_msgs, n_bytes; /* what left this rank: the tests read it through FAKE_RCCL_STATS */
} fake_comm;

typedef struct {
    int is_send;
    void *buf;
    size_t bytes;
    int peer;
    fake_comm *cm;
    hipStream_t stream;
    char *host; /* page-locked staging of this message */
} op_t;

/* one enqueued transfer: what the host function works on, alive until the stream has passed the copies behind it */
typedef struct job {
    op_t *ops;
    int n;
    fake_comm *cm;
    /* all-reduce: the staging holds world vectors, `red_bytes` apart; the host function sums them into the first */
    int red_dt; /* 0: none; else the ncclDataType_t + 1 */
    size_t red_count, red_bytes;
    char *red_host;
    char *stage;      /* one page-locked block behind every ops[i].host / red_host */
    hipEvent_t done;  /* recorded behind the job's last copy */
    struct job *next;
} job_t;

static __thread int g_depth = 0;
static __thread op_t g_ops[MAX_OPS];
static __thread int g_nops = 0;
static char g_err[256];
static pthread_mutex_t g_jobs_mu = PTHREAD_MUTEX_INITIALIZER;
static job_t *g_jobs = NULL; /* enqueued, not yet known to be through */

static int sync_mode(void)
{
    const char *e = getenv("FAKE_RCCL_SYNC");
    return e && *e == '1';
}

static size_t dt_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    case ncclFloat16: return 2;
    default: return 0;
    }
}

static int write_all(int fd, const void *p, size_t n)
{
    const char *c = (const char *)p;
    while (n) {
        ssize_t w = send(fd, c, n, MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR)
                continue;
            return -1;
        }
        c += w;
        n -= (size_t)w;
    }
    return 0;
}

static int read_all(int fd, void *p, size_t n)
{
    char *c = (char *)p;
    while (n) {
        ssize_t r = recv(fd, c, n, 0);
        if (r < 0) {
            if (errno == EINTR)
                continue;
            return -1;
        }
        if (r == 0)
            return -1; /* the peer is gone */
        c += r;
        n -= (size_t)r;
    }
    return 0;
}

static void die(const char *what)
{
    fprintf(stderr, "fake rccl: %s -- aborting (a transfer that was already enqueued cannot return an error)\n", what);
    fflush(stderr);
    abort();
}

static void *send_thread(void *arg)
{
    job_t *j = (job_t *)arg;
    for (int i = 0; i < j->n; i++) {
        op_t *o = &j->ops[i];
        if (!o->is_send || o->peer == o->cm->rank)
            continue;
        uint64_t hdr = o->bytes;
        if (write_all(o->cm->fd[o->peer], &hdr, 8) || write_all(o->cm->fd[o->peer], o->host, o->bytes))
            die("send failed (peer gone?)");
        o->cm->n_msgs++;
        o->cm->n_bytes += o->bytes;
    }
    return NULL;
}

/* the host function of a job: runs in stream order, behind the copies that filled the outgoing staging; no HIP calls */
static void transfer_fn(void *arg)
{
    job_t *j = (job_t *)arg;
    pthread_t th;
    int need_thread = 0;
    for (int i = 0; i < j->n; i++)
        if (j->ops[i].is_send && j->ops[i].peer != j->cm->rank)
            need_thread = 1;
    if (need_thread && pthread_create(&th, NULL, send_thread, j))
        die("pthread_create");
    for (int i = 0; i < j->n; i++) {
        op_t *o = &j->ops[i];
        if (o->is_send || o->peer == o->cm->rank)
            continue;
        uint64_t hdr = 0;
        char msg[200];
        if (read_all(o->cm->fd[o->peer], &hdr, 8)) {
            snprintf(msg, sizeof(msg), "rank %d lost its peer %d", o->cm->rank, o->peer);
            die(msg);
        }
        if (hdr != o->bytes) {
            snprintf(msg, sizeof(msg), "rank %d posted a recv of %zu bytes from %d, the message has %llu", o->cm->rank,
                     o->bytes, o->peer, (unsigned long long)hdr);
            die(msg);
        }
        if (read_all(o->cm->fd[o->peer], o->host, o->bytes))
            die("recv failed");
    }
    if (need_thread)
        pthread_join(th, NULL);
    if (j->red_dt) { /* all-reduce: sum the world's vectors into the first */
        const ncclDataType_t dt = (ncclDataType_t)(j->red_dt - 1);
        char *h = j->red_host;
        for (int p = 1; p < j->cm->world; p++) {
            if (dt == ncclUint8) {
                uint8_t *a = (uint8_t *)h, *q = (uint8_t *)(h + (size_t)p * j->red_bytes);
                for (size_t i = 0; i < j->red_count; i++)
                    a[i] = (uint8_t)(a[i] + q[i]); /* wraps, like the real thing */
            } else if (dt == ncclInt32) {
                int32_t *a = (int32_t *)h, *q = (int32_t *)(h + (size_t)p * j->red_bytes);
                for (size_t i = 0; i < j->red_count; i++)
                    a[i] += q[i];
            } else {
                uint64_t *a = (uint64_t *)h, *q = (uint64_t *)(h + (size_t)p * j->red_bytes);
                for (size_t i = 0; i < j->red_count; i++)
                    a[i] += q[i];
            }
        }
    }
}

/* jobs whose last copy the stream has passed are freed (called from the API entry points: HIP calls are allowed here) */
static void reap_jobs(int wait_all)
{
    pthread_mutex_lock(&g_jobs_mu);
    job_t **pp = &g_jobs;
    while (*pp) {
        job_t *j = *pp;
        hipError_t e = wait_all ? hipEventSynchronize(j->done) : hipEventQuery(j->done);
        if (e == hipSuccess) {
            *pp = j->next;
            (void)hipEventDestroy(j->done);
            if (j->stage)
                (void)hipHostFree(j->stage);
            free(j->ops);
            free(j);
        } else {
            (void)hipGetLastError();
            pp = &j->next;
        }
    }
    pthread_mutex_unlock(&g_jobs_mu);
}

/* the tail every job shares: host function, copies of what came in, the event that says the job is through */
static ncclResult_t launch_job(job_t *j, hipStream_t s)
{
    if (hipLaunchHostFunc(s, transfer_fn, j) != hipSuccess)
        return ncclUnhandledCudaError;
    for (int i = 0; i < j->n; i++) {
        op_t *o = &j->ops[i];
        if (o->is_send || o->peer == o->cm->rank || !o->bytes || !o->buf)
            continue;
        if (hipMemcpyAsync(o->buf, o->host, o->bytes, hipMemcpyHostToDevice, s) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

static ncclResult_t finish_job(job_t *j, hipStream_t s)
{
    if (hipEventCreateWithFlags(&j->done, hipEventDisableTiming) != hipSuccess || hipEventRecord(j->done, s) != hipSuccess)
        return ncclUnhandledCudaError;
    pthread_mutex_lock(&g_jobs_mu);
    j->next = g_jobs;
    g_jobs = j;
    pthread_mutex_unlock(&g_jobs_mu);
    if (sync_mode() && hipStreamSynchronize(s) != hipSuccess)
        return ncclUnhandledCudaError;
    return ncclSuccess;
}

/* enqueues the queued operations of one (outermost) group on their stream and returns */
static ncclResult_t enqueue_ops(op_t *ops_in, int n)
{
    reap_jobs(0);
    if (n == 0)
        return ncclSuccess;
    hipStream_t s = ops_in[0].stream;
    for (int i = 1; i < n; i++)
        if (ops_in[i].stream != s) {
            snprintf(g_err, sizeof(g_err), "fake rccl: one group on several streams is not supported");
            return ncclInvalidUsage;
        }
    job_t *j = (job_t *)calloc(1, sizeof(job_t));
    j->ops = (op_t *)malloc(sizeof(op_t) * (size_t)n);
    memcpy(j->ops, ops_in, sizeof(op_t) * (size_t)n);
    j->n = n;
    j->cm = ops_in[0].cm;
    op_t *ops = j->ops;
    /* self sends pair with self receives, in order: device to device, in stream order */
    int ri = 0;
    for (int i = 0; i < n; i++) {
        if (!ops[i].is_send || ops[i].peer != ops[i].cm->rank)
            continue;
        while (ri < n && !(!ops[ri].is_send && ops[ri].peer == ops[ri].cm->rank))
            ri++;
        if (ri == n || ops[ri].bytes != ops[i].bytes) {
            snprintf(g_err, sizeof(g_err), "fake rccl: send to self of %zu bytes has no matching recv", ops[i].bytes);
            return ncclInvalidUsage;
        }
        if (ops[i].bytes && hipMemcpyAsync(ops[ri].buf, ops[i].buf, ops[i].bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return ncclUnhandledCudaError;
        ri++;
    }
    /* page-locked staging for everything that crosses a socket */
    size_t total = 0;
    for (int i = 0; i < n; i++)
        if (ops[i].peer != ops[i].cm->rank)
            total += (ops[i].bytes + 63) & ~(size_t)63;
    if (total && hipHostMalloc((void **)&j->stage, total, hipHostMallocDefault) != hipSuccess)
        return ncclUnhandledCudaError;
    size_t at = 0;
    for (int i = 0; i < n; i++) {
        ops[i].host = NULL;
        if (ops[i].peer == ops[i].cm->rank)
            continue;
        ops[i].host = j->stage + at;
        at += (ops[i].bytes + 63) & ~(size_t)63;
        if (ops[i].is_send && ops[i].bytes &&
            hipMemcpyAsync(ops[i].host, ops[i].buf, ops[i].bytes, hipMemcpyDeviceToHost, s) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    ncclResult_t r = launch_job(j, s);
    if (r != ncclSuccess)
        return r;
    return finish_job(j, s);
}

static ncclResult_t post(int is_send, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    fake_comm *cm = (fake_comm *)comm;
    if (!cm || peer < 0 || peer >= cm->world || !dt_size(dt))
        return ncclInvalidArgument;
    if (g_nops == MAX_OPS)
        return ncclInternalError;
    op_t *o = &g_ops[g_nops++];
    o->is_send = is_send;
    o->buf = buf;
    o->bytes = count * dt_size(dt);
    o->peer = peer;
    o->cm = cm;
    o->stream = s;
    o->host = NULL;
    if (g_depth == 0) {
        ncclResult_t r = enqueue_ops(g_ops, g_nops);
        g_nops = 0;
        return r;
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
    g_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0)
        return ncclInvalidUsage;
    if (--g_depth)
        return ncclSuccess;
    ncclResult_t r = enqueue_ops(g_ops, g_nops);
    g_nops = 0;
    return r;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    return post(1, (void *)buf, count, dt, peer, comm, s);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s)
{
    return post(0, buf, count, dt, peer, comm, s);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t s)
{
    fake_comm *cm = (fake_comm *)comm;
    if (!cm || !dt_size(dt))
        return ncclInvalidArgument;
    if (g_depth != 0) {
        snprintf(g_err, sizeof(g_err), "fake rccl: ncclAllGather inside a group is not supported");
        return ncclInvalidUsage;
    }
    const size_t b = count * dt_size(dt);
    char *mine = (char *)recv + (size_t)cm->rank * b;
    if ((const void *)mine != send && b && hipMemcpyAsync(mine, send, b, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return ncclUnhandledCudaError;
    ncclGroupStart();
    ncclResult_t r = ncclSuccess;
    for (int p = 0; p < cm->world && r == ncclSuccess; p++) {
        if (p == cm->rank)
            continue;
        r = ncclSend(mine, count, dt, p, comm, s);
        if (r == ncclSuccess)
            r = ncclRecv((char *)recv + (size_t)p * b, count, dt, p, comm, s);
    }
    ncclResult_t e = ncclGroupEnd();
    return r != ncclSuccess ? r : e;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t s)
{
    fake_comm *cm = (fake_comm *)comm;
    if (!cm || op != ncclSum || !(dt == ncclUint8 || dt == ncclInt32 || dt == ncclUint64 || dt == ncclInt64))
        return ncclInvalidArgument;
    if (g_depth != 0)
        return ncclInvalidUsage;
    reap_jobs(0);
    const size_t b = count * dt_size(dt);
    if (!b)
        return ncclSuccess;
    /* every rank's vector to every rank, summed by the host function in page-locked memory, the sum copied to `recv`:
       one job = [D2H own vector] -> host function (exchange + sum) -> [H2D sum], all in stream order */
    job_t *j = (job_t *)calloc(1, sizeof(job_t));
    const int n = 2 * (cm->world - 1);
    j->ops = (op_t *)calloc((size_t)(n ? n : 1), sizeof(op_t));
    j->n = n;
    j->cm = cm;
    const size_t slot = (b + 63) & ~(size_t)63;
    if (hipHostMalloc((void **)&j->stage, slot * (size_t)cm->world, hipHostMallocDefault) != hipSuccess)
        return ncclUnhandledCudaError;
    /* slot 0 = this rank's own vector (the sum ends there), slots 1.. = the peers' in rank order */
    if (hipMemcpyAsync(j->stage, send, b, hipMemcpyDeviceToHost, s) != hipSuccess)
        return ncclUnhandledCudaError;
    int q = 0, sl = 1;
    for (int p = 0; p < cm->world; p++) {
        if (p == cm->rank)
            continue;
        op_t *o = &j->ops[q++];
        o->is_send = 1; o->buf = NULL; o->bytes = b; o->peer = p; o->cm = cm; o->stream = s; o->host = j->stage;
        o = &j->ops[q++];
        o->is_send = 0; o->buf = NULL; o->bytes = b; o->peer = p; o->cm = cm; o->stream = s;
        o->host = j->stage + slot * (size_t)sl++;
    }
    j->red_dt = (int)dt + 1;
    j->red_count = count;
    j->red_bytes = slot;
    j->red_host = j->stage;
    ncclResult_t r = launch_job(j, s); /* (no receive has a device buffer: nothing is copied back by it) */
    if (r != ncclSuccess)
        return r;
    if (hipMemcpyAsync(recv, j->stage, b, hipMemcpyHostToDevice, s) != hipSuccess)
        return ncclUnhandledCudaError;
    return finish_job(j, s);
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id, 0, sizeof(*id));
    const char *dir = getenv("FAKE_RCCL_DIR");
    snprintf(id->internal, 99, "%s/brxfake.%d.%llx", dir && *dir ? dir : "/tmp", (int)getpid(),
             (unsigned long long)ts.tv_nsec ^ ((unsigned long long)ts.tv_sec << 20));
    return ncclSuccess;
}

static int unix_addr(struct sockaddr_un *a, const char *base, int rank)
{
    memset(a, 0, sizeof(*a));
    a->sun_family = AF_UNIX;
    return snprintf(a->sun_path, sizeof(a->sun_path), "%s.%d", base, rank) < (int)sizeof(a->sun_path) ? 0 : -1;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank)
{
    if (!out || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world)
        return ncclInvalidArgument;
    fake_comm *cm = (fake_comm *)calloc(1, sizeof(fake_comm));
    cm->world = world;
    cm->rank = rank;
    for (int p = 0; p < MAX_WORLD; p++)
        cm->fd[p] = -1;
    cm->listen_fd = -1;
    memcpy(cm->base, id.internal, sizeof(cm->base) - 1); /* (the id is a path of at most 99 characters) */
    cm->base[sizeof(cm->base) - 1] = 0;
    struct sockaddr_un a;
    if (world > 1) {
        /* listen first, then dial the lower ranks (a connect completes against the backlog), then accept the higher ones */
        cm->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
        if (cm->listen_fd < 0 || unix_addr(&a, cm->base, rank))
            goto fail;
        unlink(a.sun_path);
        if (bind(cm->listen_fd, (struct sockaddr *)&a, sizeof(a)) || listen(cm->listen_fd, MAX_WORLD))
            goto fail;
        for (int p = 0; p < rank; p++) {
            int fd = socket(AF_UNIX, SOCK_STREAM, 0);
            if (fd < 0 || unix_addr(&a, cm->base, p))
                goto fail;
            int ok = 0;
            for (int tries = 0; tries < 1200 && !ok; tries++) { /* up to 120 s for the peer to come up */
                if (connect(fd, (struct sockaddr *)&a, sizeof(a)) == 0)
                    ok = 1;
                else
                    usleep(100000);
            }
            if (!ok) {
                close(fd);
                goto fail;
            }
            int32_t me = rank;
            if (write_all(fd, &me, 4)) {
                close(fd);
                goto fail;
            }
            cm->fd[p] = fd;
        }
        struct timeval tv = {120, 0};
        setsockopt(cm->listen_fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        for (int n = rank + 1; n < world; n++) {
            int fd = accept(cm->listen_fd, NULL, NULL);
            int32_t who = -1;
            if (fd < 0 || read_all(fd, &who, 4) || who <= rank || who >= world || cm->fd[who] >= 0) {
                if (fd >= 0)
                    close(fd);
                goto fail;
            }
            cm->fd[who] = fd;
        }
    }
    *out = (ncclComm_t)cm;
    return ncclSuccess;
fail:
    fprintf(stderr, "fake rccl: rank %d of %d could not join (%s)\n", rank, world, strerror(errno));
    ncclCommDestroy((ncclComm_t)cm);
    return ncclSystemError;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    fake_comm *cm = (fake_comm *)comm;
    if (!cm)
        return ncclSuccess;
    reap_jobs(1); /* transfers still enqueued use the sockets closed below */
    const char *st = getenv("FAKE_RCCL_STATS");
    if (st && *st) {
        char path[512];
        snprintf(path, sizeof(path), "%s.rank%d", st, cm->rank);
        FILE *f = fopen(path, "w");
        if (f) {
            fprintf(f, "%llu %llu\n", (unsigned long long)cm->n_msgs, (unsigned long long)cm->n_bytes);
            fclose(f);
        }
    }
    for (int p = 0; p < MAX_WORLD; p++)
        if (cm->fd[p] >= 0)
            close(cm->fd[p]);
    if (cm->listen_fd >= 0) {
        struct sockaddr_un a;
        close(cm->listen_fd);
        if (!unix_addr(&a, cm->base, cm->rank))
            unlink(a.sun_path);
    }
    free(cm);
    return ncclSuccess;
}

/* a rank that failed between two collectives leaves: its sockets are shut down, so that the peers' transfers end (here:
 * their host functions abort their processes, loudly) instead of waiting for bytes that will never come */
ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    fake_comm *cm = (fake_comm *)comm;
    if (!cm)
        return ncclSuccess;
    for (int p = 0; p < MAX_WORLD; p++)
        if (cm->fd[p] >= 0)
            shutdown(cm->fd[p], SHUT_RDWR);
    return ncclCommDestroy(comm);
}

const char *ncclGetErrorString(ncclResult_t r)
{
    if (g_err[0])
        return g_err;
    switch (r) {
    case ncclSuccess: return "fake rccl: success";
    case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
    case ncclSystemError: return "fake rccl: socket error";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
    }
}
