// memops_lab.hip - r04 lab (not product code): what do hipStreamWriteValue64 / hipStreamWaitValue64 on host-registered POSIX
// shared memory cost on this runtime, do they work between two PROCESSES that share one GPU, and how do they behave beside a
// kernel that occupies every CU (the L2-swept SpMM's shape: 256 workgroups x 1024 threads, 128 registers, 144 KiB LDS)?
// Compared with today's publish path (hipLaunchHostFunc storing the word) and with two copy-engine alternatives (an 8-byte
// hipMemcpyAsync out of a pinned ring; a spin kernel of ONE wave polling the word).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/memops_lab tools/memops_lab.hip -lrt && tools/memops_lab
//
// Every wait in here is released by the host after 2 s at the latest (the words live in host memory), so nothing can hang.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                       \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            printf("FAILED %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);             \
            fflush(stdout);                                                                         \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

static double now_us()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

// the swept kernel's footprint: one workgroup per CU, whole register file and LDS; spins for `us` microseconds
__global__ void __launch_bounds__(1024) hog_kernel(float *out, long long ticks)
{
    extern __shared__ float lds[];
    float acc[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) acc[i] = threadIdx.x * 0.001f + i;
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 96; ++i) acc[i] = acc[i] * 1.0001f + lds[(threadIdx.x + i) & 1023];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

__global__ void tiny_kernel(uint64_t *p, uint64_t v) { *p = v; }

// ONE wave polling a word in host memory (system scope) until it reaches `want`, bounded
__global__ void spin_kernel(const uint64_t *w, uint64_t want, long long max_ticks, uint64_t *seen)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    uint64_t v;
    do {
        v = __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v >= want) break;
        __builtin_amdgcn_s_sleep(20);
    } while (wall_clock64() - t0 < max_ticks);
    *seen = v;
}

struct Store {
    volatile uint64_t *dst;
    uint64_t val;
};
static void store_cb(void *a)
{
    Store *s = (Store *)a;
    __atomic_store_n(s->dst, s->val, __ATOMIC_RELEASE);
}

struct Block {
    volatile uint64_t *w;
    uint64_t want;
    double waited_us;
    int timed_out;
};
// a host function that BLOCKS its stream until *w >= want (bounded: 2 s) - "hipStreamWaitValue on the host"
static void block_cb(void *a)
{
    Block *b = (Block *)a;
    const double t0 = now_us();
    b->timed_out = 0;
    while (__atomic_load_n(b->w, __ATOMIC_ACQUIRE) < b->want)
        if (now_us() - t0 > 2e6) {
            b->timed_out = 1;
            break;
        }
    b->waited_us = now_us() - t0;
}

// host waits (bounded) until *w >= v; returns elapsed us or -1
static double host_wait(volatile uint64_t *w, uint64_t v, double t0, double timeout_us = 2e6)
{
    while (__atomic_load_n(w, __ATOMIC_ACQUIRE) < v)
        if (now_us() - t0 > timeout_us) return -1;
    return now_us() - t0;
}

static int run_child(const char *name);

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    char name[64];
    snprintf(name, sizeof name, "/ngcf_memops_lab_%d", (int)getpid());
    (void)argc, (void)argv;
    // the child process is forked BEFORE this process touches the GPU (no exec: each side initialises HIP by itself afterwards)
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) { printf("shm failed\n"); return 1; }
    uint64_t *shm = (uint64_t *)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    memset((void *)shm, 0, 4096);
    const bool no_child = getenv("LAB_NO_CHILD") != nullptr;      // (under a profiler: its library has initialised the GPU already)
    pid_t child = no_child ? -1 : fork();
    if (child == 0) _exit(run_child(name));

    int dev = 0, can = -1;
    CK(hipSetDevice(0));
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("device %s, CUs %d, hipDeviceAttributeCanUseStreamWaitValue = %d\n", prop.gcnArchName, prop.multiProcessorCount, can);
    CK(hipHostRegister((void *)shm, 4096, hipHostRegisterMapped | hipHostRegisterPortable));
    uint64_t *dshm = nullptr;
    CK(hipHostGetDevicePointer((void **)&dshm, (void *)shm, 0));
    printf("shm host %p device %p\n", (void *)shm, (void *)dshm);

    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    float *dout;
    uint64_t *dword, *dseen;
    CK(hipMalloc(&dout, 4096));
    CK(hipMalloc(&dword, 4096));
    CK(hipMalloc(&dseen, 4096));
    uint64_t *pinned;
    CK(hipHostMalloc((void **)&pinned, 4096, hipHostMallocDefault));
    for (int i = 0; i < 512; ++i) pinned[i] = 1000 + i;
    uint64_t *dvals;                                               // dvals[i] = 1000 + i on the device: sources of truly asynchronous 8-byte copies
    CK(hipMalloc(&dvals, 4096));
    CK(hipMemcpy(dvals, pinned, 4096, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
    printf("wall clock %d kHz\n", wall_khz);
    auto ticks_of_us = [&](double us) { return (long long)(us * 1e-3 * wall_khz); };
    volatile uint64_t *w = shm;   // words: 0..15 ours, 16..31 the child's

    // ---- A: write-value, idle GPU
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now_us();
        hipError_t e = hipStreamWriteValue64(s1, dshm + 0, 10 + rep, 0);
        double t1 = now_us();
        if (e != hipSuccess) { printf("A: hipStreamWriteValue64 failed: %s\n", hipGetErrorString(e)); break; }
        double dt = host_wait(w + 0, 10 + rep, t0);
        printf("A write-value idle GPU: enqueue %.1f us, value visible on the host after %.1f us\n", t1 - t0, dt);
    }
    // ---- A2: the alternatives, idle GPU
    for (int rep = 0; rep < 3; ++rep) {
        static Store st;
        st.dst = w + 1;
        st.val = 20 + rep;
        double t0 = now_us();
        CK(hipLaunchHostFunc(s1, store_cb, &st));
        double t1 = now_us();
        printf("A2 host function idle GPU: enqueue %.1f us, visible after %.1f us\n", t1 - t0, host_wait(w + 1, 20 + rep, t0));
        t0 = now_us();
        CK(hipMemcpyAsync(dshm + 2, pinned + rep, 8, hipMemcpyHostToHost, s1));
        t1 = now_us();
        printf("A2 8-byte copy pinned -> shm (HostToHost) idle GPU: enqueue %.1f us, visible after %.1f us\n", t1 - t0, host_wait(w + 2, 1000 + rep, t0));
        CK(hipStreamSynchronize(s1));
        CK(hipMemcpyAsync(dword, pinned + 100 + rep, 8, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s1));
        t0 = now_us();
        CK(hipMemcpyAsync(dshm + 3, dword, 8, hipMemcpyDeviceToHost, s1));
        t1 = now_us();
        printf("A2 8-byte copy device -> shm idle GPU: enqueue %.1f us, visible after %.1f us\n", t1 - t0, host_wait(w + 3, 1100 + rep, t0));
        CK(hipStreamSynchronize(s1));
    }
    // ---- B: wait-value released by the host
    for (int rep = 0; rep < 3; ++rep) {
        w[4] = 0;
        double t0 = now_us();
        hipError_t e = hipStreamWaitValue64(s1, dshm + 4, 5 + rep, hipStreamWaitValueGte, ~0ull);
        if (e != hipSuccess) { printf("B: hipStreamWaitValue64 failed: %s\n", hipGetErrorString(e)); break; }
        CK(hipMemcpyAsync(dshm + 5, dvals + 200 + rep, 8, hipMemcpyDeviceToHost, s1));     // what follows the wait (a HostToHost copy would block the host here)
        double t1 = now_us();
        usleep(3000);
        const bool early = w[5] == 1200 + rep;
        double t2 = now_us();
        __atomic_store_n(w + 4, 5 + rep, __ATOMIC_RELEASE);
        double dt = host_wait(w + 5, 1200 + rep, t2);
        if (dt < 0) { __atomic_store_n(w + 4, ~0ull >> 1, __ATOMIC_RELEASE); }
        CK(hipStreamSynchronize(s1));
        printf("B wait-value on shm: enqueue %.1f us, ran early (bug) %d, copy behind it visible %.1f us after the host's store\n", t1 - t0, (int)early, dt);
    }
    // ---- C: between processes: the child's stream writes word 16 when we store word 6; its wait releases on our write-value to word 7
    if (!no_child) {
        __atomic_store_n(w + 15, 1, __ATOMIC_RELEASE);             // "parent is ready"
        double t0 = now_us();
        if (host_wait(w + 31, 1, t0, 60e6) < 0) printf("C: the child never came up\n");
        for (int rep = 0; rep < 3; ++rep) {
            // we wait (on the GPU) for the child's write-value of 100+rep into word 16; the child issues it when it sees word 6 = rep+1
            hipError_t e = hipStreamWaitValue64(s1, dshm + 16, 100 + rep, hipStreamWaitValueGte, ~0ull);
            if (e != hipSuccess) { printf("C: wait failed %s\n", hipGetErrorString(e)); break; }
            CK(hipMemcpyAsync(dshm + 8, dvals + 300 + rep, 8, hipMemcpyDeviceToHost, s1));
            usleep(1000);
            t0 = now_us();
            __atomic_store_n(w + 6, rep + 1, __ATOMIC_RELEASE);
            double dt = host_wait(w + 8, 1300 + rep, t0);
            if (dt < 0) __atomic_store_n(w + 16, ~0ull >> 1, __ATOMIC_RELEASE);
            CK(hipStreamSynchronize(s1));
            printf("C cross-process: child write-value -> our wait-value -> copy visible %.1f us after the go signal (incl. the child's host latency)\n", dt);
        }
    }
    // ---- D: beside a kernel that holds every CU for 3 ms
    const long long hog_ticks = ticks_of_us(3000);
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            w[9] = 0;
            w[10] = 0;
            CK(hipDeviceSynchronize());
            hog_kernel<<<prop.multiProcessorCount, 1024, 144 * 1024, s2>>>(dout, hog_ticks);
            usleep(300);                                            // the hog is resident
            static Store st;
            double t0 = now_us();
            const uint64_t v = 1050 + rep;
            const char *what = "";
            if (mode == 0) {
                what = "write-value";
                CK(hipStreamWriteValue64(s1, dshm + 9, v, 0));
            } else if (mode == 1) {
                what = "host function";
                st.dst = w + 9;
                st.val = v;
                CK(hipLaunchHostFunc(s1, store_cb, &st));
            } else if (mode == 2) {
                what = "8-byte copy device -> shm";
                CK(hipMemcpyAsync(dshm + 9, dvals + (v - 1000), 8, hipMemcpyDeviceToHost, s1));
            } else if (mode == 3) {
                what = "tiny kernel storing the word";
                tiny_kernel<<<1, 1, 0, s1>>>(dshm + 9, v);
            } else if (mode == 4) {
                what = "wait-value (released at once by the host) + 8-byte copy device -> shm";
                CK(hipStreamWaitValue64(s1, dshm + 10, 1, hipStreamWaitValueGte, ~0ull));
                CK(hipMemcpyAsync(dshm + 9, dvals + (v - 1000), 8, hipMemcpyDeviceToHost, s1));
                __atomic_store_n(w + 10, 1, __ATOMIC_RELEASE);
            } else {
                what = "one-wave spin kernel on the word (released at once) + tiny kernel";
                spin_kernel<<<1, 64, 0, s1>>>(dshm + 10, 1, ticks_of_us(1e6), dseen);
                tiny_kernel<<<1, 1, 0, s1>>>(dshm + 9, v);
                __atomic_store_n(w + 10, 1, __ATOMIC_RELEASE);
            }
            double dt = host_wait(w + 9, v, t0);
            if (dt < 0) __atomic_store_n(w + 10, ~0ull >> 1, __ATOMIC_RELEASE);
            double t1 = now_us();
            CK(hipStreamSynchronize(s2));
            double hog_end = now_us();
            CK(hipDeviceSynchronize());
            printf("D beside the CU hog: %-68s visible after %8.1f us (hog ended %.1f us after that)\n", what, dt, hog_end - t1);
        }
    }
    // ---- E: does a pending wait-value occupy a CU?  hog launched AFTER an unsatisfied wait; does the hog still get every CU at once?
    for (int mode = 0; mode < 2; ++mode) {
        w[11] = 0;
        CK(hipDeviceSynchronize());
        if (mode == 0)
            CK(hipStreamWaitValue64(s1, dshm + 11, 1, hipStreamWaitValueGte, ~0ull));
        else
            spin_kernel<<<1, 64, 0, s1>>>(dshm + 11, 1, ticks_of_us(1e6), dseen);
        usleep(500);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s2));
        hog_kernel<<<prop.multiProcessorCount, 1024, 144 * 1024, s2>>>(dout, ticks_of_us(1000));
        CK(hipEventRecord(e1, s2));
        double t0 = now_us();
        hipError_t q;
        while ((q = hipEventQuery(e1)) == hipErrorNotReady && now_us() - t0 < 50000) {}
        const bool done = q == hipSuccess;
        __atomic_store_n(w + 11, 1, __ATOMIC_RELEASE);
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("E 1 ms hog launched behind a pending %s: finished before the release %d, took %.3f ms (1.0 = every CU at once, 2.0 = one CU was taken)\n",
               mode == 0 ? "wait-value" : "spin kernel", (int)done, ms);
    }
    // ---- F: a BLOCKING host function on a copy stream as the wait (the main thread never blocks).  Do the callbacks of different
    // streams run on different threads - i.e. can a store callback on one stream release a blocked callback on another?
    {
        hipStream_t st[8];
        for (int i = 0; i < 8; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        static Block blk[8];
        static Store rel;
        for (int n_blocked : {1, 7}) {
            w[12] = 0;
            for (int i = 0; i < n_blocked; ++i) {
                blk[i].w = w + 12;
                blk[i].want = 1;
                CK(hipLaunchHostFunc(st[i], block_cb, &blk[i]));
                CK(hipMemcpyAsync(dshm + 40 + i, dvals + i, 8, hipMemcpyDeviceToHost, st[i]));     // what the wait guards
            }
            usleep(2000);                                           // the callbacks are running (blocked)
            rel.dst = w + 12;
            rel.val = 1;
            double t0 = now_us();
            CK(hipLaunchHostFunc(st[7], store_cb, &rel));           // the release arrives as a host function of ANOTHER stream
            for (int i = 0; i < n_blocked; ++i) CK(hipStreamSynchronize(st[i]));
            double dt = now_us() - t0;
            int to = 0;
            for (int i = 0; i < n_blocked; ++i) to += blk[i].timed_out;
            printf("F %d stream(s) blocked in a host function, released by a host function on another stream: all done %.1f us after the release was "
                   "enqueued, timed out %d (0 = callbacks of different streams do not serialise)\n", n_blocked, dt, to);
            CK(hipDeviceSynchronize());
        }
        // the wait inside a callback vs the main thread waiting and then enqueueing: time from the release to the guarded copy's completion
        for (int rep = 0; rep < 3; ++rep) {
            w[12] = 0;
            w[50] = 0;
            blk[0].w = w + 12;
            blk[0].want = 1;
            CK(hipLaunchHostFunc(st[0], block_cb, &blk[0]));
            CK(hipMemcpyAsync(dshm + 50, dvals + 60 + rep, 8, hipMemcpyDeviceToHost, st[0]));
            usleep(1000);
            double t0 = now_us();
            __atomic_store_n(w + 12, 1, __ATOMIC_RELEASE);
            double dt = host_wait(w + 50, 1060 + rep, t0);
            CK(hipStreamSynchronize(st[0]));
            w[12] = 0;
            w[51] = 0;
            usleep(1000);
            double t1 = now_us();
            __atomic_store_n(w + 12, 1, __ATOMIC_RELEASE);
            host_wait(w + 12, 1, t1);                                // (the main thread sees the word ...)
            CK(hipMemcpyAsync(dshm + 51, dvals + 70 + rep, 8, hipMemcpyDeviceToHost, st[1]));      // ... and only then enqueues the copy
            double dt2 = host_wait(w + 51, 1070 + rep, t1);
            CK(hipStreamSynchronize(st[1]));
            printf("F release -> guarded 8-byte copy done: %.1f us with the wait inside a host function on the copy stream, %.1f us with the main "
                   "thread waiting and then enqueueing\n", dt, dt2);
        }
    }
    __atomic_store_n(w + 14, 1, __ATOMIC_RELEASE);                 // tell the child to leave
    int st = 0;
    if (!no_child) {
        waitpid(child, &st, 0);
        printf("child exit %d\n", WEXITSTATUS(st));
    }
    CK(hipHostUnregister((void *)shm));
    shm_unlink(name);
    printf("done\n");
    return 0;
}

static int run_child(const char *name)
{
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return 2;
    uint64_t *shm = (uint64_t *)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    volatile uint64_t *w = shm;
    double t0 = now_us();
    if (host_wait(w + 15, 1, t0, 60e6) < 0) return 3;
    CK(hipSetDevice(0));
    CK(hipHostRegister((void *)shm, 4096, hipHostRegisterMapped | hipHostRegisterPortable));
    uint64_t *dshm = nullptr;
    CK(hipHostGetDevicePointer((void **)&dshm, (void *)shm, 0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    __atomic_store_n(w + 31, 1, __ATOMIC_RELEASE);
    for (int rep = 0; rep < 3; ++rep) {
        t0 = now_us();
        if (host_wait(w + 6, rep + 1, t0, 10e6) < 0) break;
        hipError_t e = hipStreamWriteValue64(s, dshm + 16, 100 + rep, 0);
        if (e != hipSuccess) printf("child: write-value failed %s\n", hipGetErrorString(e));
    }
    t0 = now_us();
    host_wait(w + 14, 1, t0, 60e6);
    CK(hipDeviceSynchronize());
    CK(hipHostUnregister((void *)shm));
    return 0;
}
