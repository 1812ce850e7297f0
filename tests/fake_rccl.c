/* TEST-ONLY transport with librccl's symbols: lets `world` PROCESSES THAT SHARE ONE GPU run the shipped exchange
 * (br_amd/csrc/brx_exchange.hip) with world > 1.  RCCL itself cannot put two ranks on one device, and the boxes this
 * repository reaches have one GPU; libbrx resolves librccl through dlopen (BRX_RCCL_PATH first), so a test selects
 * this library instead and every line of the exchange above the ten nccl* entry points runs unchanged.
 *
 * Semantics kept from RCCL, because the exchange relies on them:
 *   - ncclSend / ncclRecv between a pair of ranks match in issue order; inside ncclGroupStart/End they are deferred
 *     and executed together at the outermost GroupEnd (sends on a helper thread, receives on the caller: paired
 *     send/recv cannot deadlock whatever the socket buffers hold);
 *   - send to self pairs with the recv from self of the same group;
 *   - ncclAllGather works in place (sendbuff == recvbuff + rank * count);
 *   - ncclSum on ncclUint8 WRAPS (the exactness trick of brx_exchange_reduce_counts must hold against that).
 * Differences: everything is synchronous (the stream is drained first, results are complete on return) and bytes
 * travel device -> host -> AF_UNIX socket -> host -> device.  A message whose size differs from what the receiver
 * posted is an error here (RCCL would hang or corrupt): the test then fails loudly.
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
    void *host; /* staging */
} op_t;

static __thread int g_depth = 0;
static __thread op_t g_ops[MAX_OPS];
static __thread int g_nops = 0;
static __thread char g_err[256];

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

typedef struct {
    op_t *ops;
    int n;
    int failed;
} send_job;

static void *send_thread(void *arg)
{
    send_job *j = (send_job *)arg;
    for (int i = 0; i < j->n; i++) {
        op_t *o = &j->ops[i];
        if (!o->is_send || o->peer == o->cm->rank)
            continue;
        uint64_t hdr = o->bytes;
        if (write_all(o->cm->fd[o->peer], &hdr, 8) || write_all(o->cm->fd[o->peer], o->host, o->bytes)) {
            j->failed = 1;
            return NULL;
        }
        o->cm->n_msgs++;
        o->cm->n_bytes += o->bytes;
    }
    return NULL;
}

/* runs the queued operations of one (outermost) group */
static ncclResult_t run_ops(op_t *ops, int n)
{
    ncclResult_t rc = ncclSuccess;
    for (int i = 0; i < n; i++)
        if (hipStreamSynchronize(ops[i].stream) != hipSuccess)
            return ncclUnhandledCudaError;
    /* self sends pair with self receives, in order */
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
        if (ops[i].bytes && hipMemcpy(ops[ri].buf, ops[i].buf, ops[i].bytes, hipMemcpyDeviceToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
        ri++;
    }
    /* stage the outgoing messages on the host */
    for (int i = 0; i < n; i++) {
        ops[i].host = NULL;
        if (ops[i].peer == ops[i].cm->rank)
            continue;
        ops[i].host = malloc(ops[i].bytes ? ops[i].bytes : 1);
        if (!ops[i].host)
            return ncclSystemError;
        if (ops[i].is_send && ops[i].bytes &&
            hipMemcpy(ops[i].host, ops[i].buf, ops[i].bytes, hipMemcpyDeviceToHost) != hipSuccess)
            rc = ncclUnhandledCudaError;
    }
    send_job job = {ops, n, 0};
    pthread_t th;
    int have_thread = 0;
    if (rc == ncclSuccess) {
        if (pthread_create(&th, NULL, send_thread, &job))
            rc = ncclSystemError;
        else
            have_thread = 1;
    }
    for (int i = 0; i < n && rc == ncclSuccess; i++) {
        op_t *o = &ops[i];
        if (o->is_send || o->peer == o->cm->rank)
            continue;
        uint64_t hdr = 0;
        if (read_all(o->cm->fd[o->peer], &hdr, 8)) {
            snprintf(g_err, sizeof(g_err), "fake rccl: rank %d lost its peer %d", o->cm->rank, o->peer);
            rc = ncclSystemError;
            break;
        }
        if (hdr != o->bytes) {
            snprintf(g_err, sizeof(g_err), "fake rccl: rank %d posted a recv of %zu bytes from %d, the message has %llu",
                     o->cm->rank, o->bytes, o->peer, (unsigned long long)hdr);
            fprintf(stderr, "%s\n", g_err);
            rc = ncclInvalidUsage;
            break;
        }
        if (read_all(o->cm->fd[o->peer], o->host, o->bytes)) {
            rc = ncclSystemError;
            break;
        }
        if (o->bytes && hipMemcpy(o->buf, o->host, o->bytes, hipMemcpyHostToDevice) != hipSuccess)
            rc = ncclUnhandledCudaError;
    }
    if (have_thread) {
        if (rc != ncclSuccess) /* unblock a sender stuck on a peer that will never read */
            for (int i = 0; i < n; i++)
                if (ops[i].is_send && ops[i].peer != ops[i].cm->rank)
                    shutdown(ops[i].cm->fd[ops[i].peer], SHUT_RDWR);
        pthread_join(th, NULL);
        if (job.failed && rc == ncclSuccess)
            rc = ncclSystemError;
    }
    for (int i = 0; i < n; i++)
        free(ops[i].host);
    return rc;
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
        ncclResult_t r = run_ops(g_ops, g_nops);
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
    ncclResult_t r = run_ops(g_ops, g_nops);
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
    const size_t b = count * dt_size(dt);
    if (hipStreamSynchronize(s) != hipSuccess)
        return ncclUnhandledCudaError;
    char *mine = (char *)recv + (size_t)cm->rank * b;
    if ((const void *)mine != send && b && hipMemcpy(mine, send, b, hipMemcpyDeviceToDevice) != hipSuccess)
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
    const size_t b = count * dt_size(dt);
    /* every rank's vector to every rank through a device scratch area, summed on the host */
    char *scratch = NULL;
    if (hipMalloc((void **)&scratch, (size_t)cm->world * (b ? b : 1)) != hipSuccess)
        return ncclUnhandledCudaError;
    ncclResult_t r = ncclAllGather(send, scratch, count, dt, comm, s);
    char *h = (char *)malloc((size_t)cm->world * (b ? b : 1));
    if (r == ncclSuccess && (!h || hipMemcpy(h, scratch, (size_t)cm->world * b, hipMemcpyDeviceToHost) != hipSuccess))
        r = ncclUnhandledCudaError;
    if (r == ncclSuccess) {
        for (int p = 1; p < cm->world; p++) {
            if (dt == ncclUint8) {
                uint8_t *a = (uint8_t *)h, *q = (uint8_t *)(h + (size_t)p * b);
                for (size_t i = 0; i < count; i++)
                    a[i] = (uint8_t)(a[i] + q[i]); /* wraps, like the real thing */
            } else if (dt == ncclInt32) {
                int32_t *a = (int32_t *)h, *q = (int32_t *)(h + (size_t)p * b);
                for (size_t i = 0; i < count; i++)
                    a[i] += q[i];
            } else {
                uint64_t *a = (uint64_t *)h, *q = (uint64_t *)(h + (size_t)p * b);
                for (size_t i = 0; i < count; i++)
                    a[i] += q[i];
            }
        }
        if (b && hipMemcpy(recv, h, b, hipMemcpyHostToDevice) != hipSuccess)
            r = ncclUnhandledCudaError;
    }
    free(h);
    (void)hipFree(scratch);
    return r;
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
