// vmmprobe.hip -- can two processes share device memory through the virtual-memory-management API instead of hipIpc*?
// (round 3; companion of tools/ipc_probe.py, which found that hipIpcOpenMemHandle never returns for allocations > 2 GiB.)
// Parent: hipMemCreate (physical allocation, exportable as a POSIX file descriptor) -> export -> fd to the child over a
// socketpair (SCM_RIGHTS) -> maps it itself, fills it with a pattern.  Child: import -> reserve a VA range -> map -> set access
// -> checks the pattern, writes its own, the parent checks that.  Sizes: 1 GiB, 3 GiB, and a 3-GiB RANGE built from three 1-GiB
// physical chunks mapped back to back (what a join-free gather destination would be).  Every step is under alarm(): a step
// that hangs ends the process with a message instead of the box.
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static const char* g_step = "?";
static void on_alarm(int) {
    fprintf(stderr, "TIMEOUT in step: %s (pid %d)\n", g_step, (int)getpid());
    _exit(3);
}
#define STEP(name, expr)                                                                      \
    do {                                                                                      \
        g_step = name;                                                                        \
        alarm(20);                                                                            \
        hipError_t e_ = (expr);                                                               \
        alarm(0);                                                                             \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s failed: %s (pid %d)\n", name, hipGetErrorString(e_), (int)getpid()); \
            _exit(2);                                                                         \
        }                                                                                     \
    } while (0)

static void send_fd(int sock, int fd) {
    char b = 'x';
    iovec io{&b, 1};
    char ctl[CMSG_SPACE(sizeof(int))];
    memset(ctl, 0, sizeof(ctl));
    msghdr msg{};
    msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctl; msg.msg_controllen = sizeof(ctl);
    cmsghdr* c = CMSG_FIRSTHDR(&msg);
    c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(c), &fd, sizeof(int));
    if (sendmsg(sock, &msg, 0) < 0) { perror("sendmsg"); _exit(4); }
}
static int recv_fd(int sock) {
    char b;
    iovec io{&b, 1};
    char ctl[CMSG_SPACE(sizeof(int))];
    msghdr msg{};
    msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctl; msg.msg_controllen = sizeof(ctl);
    if (recvmsg(sock, &msg, 0) < 0) { perror("recvmsg"); _exit(4); }
    cmsghdr* c = CMSG_FIRSTHDR(&msg);
    int fd = -1;
    memcpy(&fd, CMSG_DATA(c), sizeof(int));
    return fd;
}
static void sync_byte(int sock, bool send) {
    char b = 's';
    if (send) { if (write(sock, &b, 1) != 1) _exit(5); }
    else { g_step = "waiting for the peer"; alarm(60); if (read(sock, &b, 1) != 1) _exit(5); alarm(0); }
}

__global__ void k_fill(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i * 2654435761u + seed;
}
__global__ void k_check(const unsigned* p, size_t n, unsigned seed, unsigned long long* bad) {
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != (unsigned)i * 2654435761u + seed;
    if (b) atomicAdd(bad, b);
}
static unsigned long long check(const void* p, size_t bytes, unsigned seed) {
    unsigned long long *d, h = 0;
    STEP("hipMalloc(counter)", hipMalloc(&d, 8));
    STEP("hipMemset", hipMemset(d, 0, 8));
    hipLaunchKernelGGL(k_check, dim3(2048), dim3(256), 0, 0, (const unsigned*)p, bytes / 4, seed, d);
    STEP("check kernel", hipDeviceSynchronize());
    STEP("hipMemcpy", hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return h;
}

// one case: nchunk physical allocations of chunk bytes each, mapped back to back in both processes
static int run_case(size_t chunk, int nchunk) {
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) { perror("socketpair"); return 1; }
    const size_t total = chunk * nchunk;
    pid_t pid = fork();
    const bool parent = pid != 0;
    const int sock = sv[parent ? 0 : 1];
    signal(SIGALRM, on_alarm);
    STEP("hipSetDevice", hipSetDevice(0));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    STEP("hipMemGetAllocationGranularity", hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (chunk % gran) { fprintf(stderr, "chunk not a multiple of the granularity %zu\n", gran); _exit(6); }
    std::vector<hipMemGenericAllocationHandle_t> h(nchunk);
    if (parent) {
        for (int i = 0; i < nchunk; ++i) {
            STEP("hipMemCreate", hipMemCreate(&h[i], chunk, &prop, 0));
            int fd = -1;
            STEP("hipMemExportToShareableHandle", hipMemExportToShareableHandle(&fd, h[i], hipMemHandleTypePosixFileDescriptor, 0));
            send_fd(sock, fd);
            close(fd);
        }
    } else {
        for (int i = 0; i < nchunk; ++i) {
            const int fd = recv_fd(sock);
            STEP("hipMemImportFromShareableHandle", hipMemImportFromShareableHandle(&h[i], (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
            close(fd);
        }
    }
    void* va = nullptr;
    STEP("hipMemAddressReserve", hipMemAddressReserve(&va, total, 0, nullptr, 0));
    for (int i = 0; i < nchunk; ++i) STEP("hipMemMap", hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, h[i], 0));
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    STEP("hipMemSetAccess", hipMemSetAccess(va, total, &acc, 1));
    int rc = 0;
    if (parent) {
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)va, total / 4, 17u);
        STEP("fill kernel", hipDeviceSynchronize());
        sync_byte(sock, true);            // pattern 17 is in place
        sync_byte(sock, false);           // the child has checked it and written pattern 99
        const unsigned long long bad = check(va, total, 99u);
        printf("  parent sees the child's writes: %s (%llu words differ)\n", bad ? "NO" : "yes", bad);
        rc = bad != 0;
        int status = 0;
        waitpid(pid, &status, 0);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { printf("  child exit status %d\n", WIFEXITED(status) ? WEXITSTATUS(status) : -1); rc = 1; }
    } else {
        sync_byte(sock, false);
        const unsigned long long bad = check(va, total, 17u);
        printf("  child sees the parent's writes: %s (%llu words differ)\n", bad ? "NO" : "yes", bad);
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)va, total / 4, 99u);
        STEP("fill kernel (child)", hipDeviceSynchronize());
        sync_byte(sock, true);
        rc = bad != 0;
    }
    for (int i = 0; i < nchunk; ++i) STEP("hipMemUnmap", hipMemUnmap((char*)va + (size_t)i * chunk, chunk));
    STEP("hipMemAddressFree", hipMemAddressFree(va, total));
    for (int i = 0; i < nchunk; ++i) STEP("hipMemRelease", hipMemRelease(h[i]));
    fflush(stdout);
    if (!parent) _exit(rc);
    close(sv[0]); close(sv[1]);
    return rc;
}

int main() {
    struct { size_t chunk; int n; const char* what; } cases[] = {
        {1ull << 30, 1, "one 1-GiB allocation"},
        {3ull << 30, 1, "one 3-GiB allocation (hipIpcOpenMemHandle hangs beyond 2 GiB)"},
        {1ull << 30, 3, "a 3-GiB range of three 1-GiB allocations mapped back to back"},
    };
    int rc = 0;
    for (auto& c : cases) {
        printf("%s:\n", c.what);
        fflush(stdout);
        // each case in its own process pair: the parent of the pair is a child of this one (no HIP state here before fork)
        pid_t pid = fork();
        if (pid == 0) _exit(run_case(c.chunk, c.n));
        int status = 0;
        waitpid(pid, &status, 0);
        const int r = WIFEXITED(status) ? WEXITSTATUS(status) : -1;
        printf("  -> %s\n", r == 0 ? "works" : "FAILED");
        rc |= r != 0;
    }
    return rc;
}
