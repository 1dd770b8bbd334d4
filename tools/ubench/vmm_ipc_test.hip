// Two processes on one GPU share physical allocations through the HIP virtual-memory API (hipMemCreate +
// hipMemExportToShareableHandle -> POSIX fd over a Unix socket -> hipMemImportFromShareableHandle + hipMemMap): the
// mechanism of the partitioned sharded mode (each rank owns the physical memory of its owners' sub-tables, every rank
// maps all of them into one contiguous address range).  Checks: export / import / map work on this driver, both
// processes see one layout, writes of the owner are visible to the peer after a kernel boundary, remap after "growth".
// build: hipcc --offload-arch=gfx950 -O2 -o vmm_ipc_test vmm_ipc_test.hip ; run: ./vmm_ipc_test
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s:%d %s -> %s\n", g_rank, __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
static int g_rank = -1;

static int send_fd(int sock, int fd) {
  char b = 'f';
  struct iovec io = {&b, 1};
  char ctl[CMSG_SPACE(sizeof(int))];
  memset(ctl, 0, sizeof(ctl));
  struct msghdr m = {};
  m.msg_iov = &io; m.msg_iovlen = 1; m.msg_control = ctl; m.msg_controllen = sizeof(ctl);
  struct cmsghdr *c = CMSG_FIRSTHDR(&m);
  c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
  memcpy(CMSG_DATA(c), &fd, sizeof(int));
  return sendmsg(sock, &m, 0) == 1 ? 0 : -1;
}
static int recv_fd(int sock) {
  char b;
  struct iovec io = {&b, 1};
  char ctl[CMSG_SPACE(sizeof(int))];
  struct msghdr m = {};
  m.msg_iov = &io; m.msg_iovlen = 1; m.msg_control = ctl; m.msg_controllen = sizeof(ctl);
  if (recvmsg(sock, &m, 0) != 1) return -1;
  struct cmsghdr *c = CMSG_FIRSTHDR(&m);
  if (!c || c->cmsg_type != SCM_RIGHTS) return -1;
  int fd;
  memcpy(&fd, CMSG_DATA(c), sizeof(int));
  return fd;
}
static void barrier(int sock) { char x = 'b'; (void)!write(sock, &x, 1); (void)!read(sock, &x, 1); }

__global__ void k_fill(unsigned long long *p, size_t n, unsigned long long tag) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag + i;
}
__global__ void k_check(const unsigned long long *p, size_t n, unsigned long long tag, unsigned *bad) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != tag + i) atomicAdd(bad, 1u);
}
// random probes (the real access pattern): latency of dependent loads through the mapping
__global__ void k_chase(const unsigned long long *p, size_t n, unsigned steps, unsigned long long *out) {
  unsigned long long x = threadIdx.x * 7919ull + blockIdx.x * 104729ull, acc = 0;
  for (unsigned s = 0; s < steps; ++s) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    unsigned long long v = p[(x >> 20) % n];
    acc += v;
    x ^= v;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int run(int rank, int sock) {
  g_rank = rank;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  if (rank == 0) printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
  for (int round = 0; round < 2; ++round) {   // round 1 = "growth": a new, larger range; the old one is unmapped and released
    const size_t chunk = (round ? 64u : 8u) * grec, n_sub = 6, total = chunk * n_sub;   // sub-table s belongs to rank s % 2
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, grec, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n_sub);
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t s = 0; s < n_sub; ++s) {
      if ((int)(s % 2) == rank) {
        CK(hipMemCreate(&h[s], chunk, &prop, 0));
        int fd = -1;
        CK(hipMemExportToShareableHandle(&fd, h[s], hipMemHandleTypePosixFileDescriptor, 0));
        if (send_fd(sock, fd)) { fprintf(stderr, "[%d] send_fd failed\n", rank); return 3; }
        close(fd);
      }
    }
    for (size_t s = 0; s < n_sub; ++s) {
      if ((int)(s % 2) != rank) {
        int fd = recv_fd(sock);
        if (fd < 0) { fprintf(stderr, "[%d] recv_fd failed\n", rank); return 3; }
        CK(hipMemImportFromShareableHandle(&h[s], (void *)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
        close(fd);
      }
      CK(hipMemMap((char *)va + s * chunk, chunk, 0, h[s], 0));
    }
    CK(hipMemSetAccess(va, total, &acc, 1));
    const size_t nw = chunk / 8;
    for (size_t s = 0; s < n_sub; ++s)
      if ((int)(s % 2) == rank) {
        CK(hipMemsetAsync((char *)va + s * chunk, 0, chunk, 0));
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (unsigned long long *)((char *)va + s * chunk), nw, (unsigned long long)(round * 1000 + s) << 40);
      }
    CK(hipDeviceSynchronize());
    barrier(sock);
    unsigned *bad = nullptr;
    CK(hipMalloc(&bad, 4));
    CK(hipMemset(bad, 0, 4));
    for (size_t s = 0; s < n_sub; ++s)
      hipLaunchKernelGGL(k_check, dim3(256), dim3(256), 0, 0, (const unsigned long long *)((char *)va + s * chunk), nw, (unsigned long long)(round * 1000 + s) << 40, bad);
    unsigned hb = 1;
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("[%d] round %d: %zu sub-tables of %zu MiB in one range at %p: %u wrong words\n", rank, round, n_sub, chunk >> 20, va, hb);
    if (hb) return 4;
    // dependent random loads over the whole range (own + imported chunks) vs a plain hipMalloc of the same size
    unsigned long long *out = nullptr, *plain = nullptr;
    CK(hipMalloc(&out, 64 * 64 * 8));
    CK(hipMalloc(&plain, total));
    CK(hipMemset(plain, 1, total));
    for (int which = 0; which < 2; ++which) {
      const unsigned long long *p = which ? plain : (const unsigned long long *)va;
      hipLaunchKernelGGL(k_chase, dim3(64), dim3(64), 0, 0, p, total / 8, 200u, out);
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_chase, dim3(64), dim3(64), 0, 0, p, total / 8, 2000u, out);
      CK(hipDeviceSynchronize());
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("[%d] round %d: dependent random loads through %s: %.0f ns per step\n", rank, round, which ? "hipMalloc" : "the mapped range", us * 1000 / 2000);
    }
    CK(hipFree(out)); CK(hipFree(plain)); CK(hipFree(bad));
    barrier(sock);
    CK(hipMemUnmap(va, total));
    for (size_t s = 0; s < n_sub; ++s) CK(hipMemRelease(h[s]));
    CK(hipMemAddressFree(va, total));
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    printf("[%d] round %d released; free %.1f GiB of %.1f\n", rank, round, fr / 1073741824.0, tot / 1073741824.0);
    barrier(sock);
  }
  printf("[%d] VMM_IPC_OK\n", rank);
  return 0;
}

int main() {
  int sv[2];
  if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 1; }
  pid_t pid = fork();   // before anything touches the GPU
  if (pid == 0) { close(sv[0]); _exit(run(1, sv[1])); }
  close(sv[1]);
  int rc = run(0, sv[0]);
  int st = 0;
  waitpid(pid, &st, 0);
  return rc ? rc : (WIFEXITED(st) ? WEXITSTATUS(st) : 9);
}
