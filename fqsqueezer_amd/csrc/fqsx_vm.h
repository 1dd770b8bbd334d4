// fqsx_vm.h -- host-side plumbing of the partitioned sharded mode (SURVEY.md 8e option (i); north star: "shards the key
// space by prefix across the 8 GPUs"): every rank holds the physical memory of the sub-tables its workers own and maps
// the other ranks' sub-tables next to them, so that all ranks see ONE table at `slots + sub * stride` -- the layout the
// kernels already use (the reference's counterpart: one CHT_kmer shared by all threads, fqs/application.h:51-54, owner
// functions fqs/dna.cpp:825, :2381-2386).  Look-ups of another rank's sub-table are loads over xGMI through the mapping;
// writes stay owner-only (insert phase), and a phase's collectives order them before the next phase's look-ups.
//
//   physical memory   hipMemCreate (one chunk per own sub-table), exported as a POSIX file descriptor
//   hand-over         descriptors travel over Unix-domain sockets (SCM_RIGHTS) between the ranks of the node: a full mesh
//                     set up once from names exchanged through the codec's transport (fqsx_comm all-gather)
//   one address range hipMemAddressReserve + hipMemMap of own and imported chunks + hipMemSetAccess
//
// The emulation build (tests/emu, host "device memory") does the same with memfd_create / mmap(MAP_FIXED), so the gloo
// tests run the whole protocol, descriptor passing included.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>
#include <unistd.h>
#include <cerrno>
#include <cstddef>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#ifndef FQSX_EMU
#include <hip/hip_runtime.h>
#endif

namespace fqsx_vm {

#ifndef FQSX_EMU
typedef hipMemGenericAllocationHandle_t Handle;
#else
typedef int Handle;   // the memfd of the chunk
#endif

inline std::string errno_str(const char *what) { return std::string(what) + ": " + strerror(errno); }

#ifndef FQSX_EMU
#define FQSX_VMCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return 1; } } while (0)
inline hipMemAllocationProp vm_prop(int device) {
  hipMemAllocationProp p = {};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = device;
  p.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
  return p;
}
inline int granularity(int device, uint64_t *g, std::string &err) {
  hipMemAllocationProp p = vm_prop(device);
  size_t v = 0;
  FQSX_VMCHK(hipMemGetAllocationGranularity(&v, &p, hipMemAllocationGranularityRecommended));
  *g = v;
  return 0;
}
inline int reserve(uint64_t bytes, uint64_t align, uint8_t **va, std::string &err) {
  void *p = nullptr;
  FQSX_VMCHK(hipMemAddressReserve(&p, bytes, align, nullptr, 0));
  *va = (uint8_t *)p;
  return 0;
}
inline int create(int device, uint64_t bytes, Handle *h, std::string &err) {
  hipMemAllocationProp p = vm_prop(device);
  FQSX_VMCHK(hipMemCreate(h, bytes, &p, 0));
  return 0;
}
inline int export_fd(Handle h, int *fd, std::string &err) {
  FQSX_VMCHK(hipMemExportToShareableHandle(fd, h, hipMemHandleTypePosixFileDescriptor, 0));
  return 0;
}
// (the descriptor stays the caller's to close)  HIP runtimes differ in what `osHandle` is: older ones (the ROCm 7.0 runtime a
// PyTorch wheel brings into the process) read the descriptor THROUGH the pointer, newer ones (ROCm 7.2) take the descriptor's
// value in the pointer, like CUDA.  The pointer form is tried first: a runtime of the second kind sees a number that is no
// open descriptor and returns an error, whereas the value form would make a runtime of the first kind read address `fd`.
inline int import_fd(int fd, Handle *h, std::string &err) {
  int slot = fd;
  if (hipMemImportFromShareableHandle(h, (void *)&slot, hipMemHandleTypePosixFileDescriptor) == hipSuccess) return 0;
  (void)hipGetLastError();
  FQSX_VMCHK(hipMemImportFromShareableHandle(h, (void *)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
  return 0;
}
inline int map(int device, uint8_t *va, uint64_t bytes, Handle h, std::string &err) {
  FQSX_VMCHK(hipMemMap(va, bytes, 0, h, 0));
  hipMemAccessDesc a = {};
  a.location.type = hipMemLocationTypeDevice;
  a.location.id = device;
  a.flags = hipMemAccessFlagsProtReadWrite;
  FQSX_VMCHK(hipMemSetAccess(va, bytes, &a, 1));
  return 0;
}
inline int unmap(uint8_t *va, uint64_t bytes, std::string &err) { FQSX_VMCHK(hipMemUnmap(va, bytes)); return 0; }
inline int release(Handle h, std::string &err) { FQSX_VMCHK(hipMemRelease(h)); return 0; }
inline int unreserve(uint8_t *va, uint64_t bytes, std::string &err) { FQSX_VMCHK(hipMemAddressFree(va, bytes)); return 0; }
#else
inline int granularity(int, uint64_t *g, std::string &) { *g = (uint64_t)sysconf(_SC_PAGESIZE); return 0; }
inline int reserve(uint64_t bytes, uint64_t, uint8_t **va, std::string &err) {
  void *p = mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) { err = errno_str("mmap (reserve)"); return 1; }
  *va = (uint8_t *)p;
  return 0;
}
inline int create(int, uint64_t bytes, Handle *h, std::string &err) {
  int fd = memfd_create("fqsx-subtable", MFD_CLOEXEC);
  if (fd < 0) { err = errno_str("memfd_create"); return 1; }
  if (ftruncate(fd, (off_t)bytes)) { err = errno_str("ftruncate"); close(fd); return 1; }
  *h = fd;
  return 0;
}
inline int export_fd(Handle h, int *fd, std::string &err) {
  *fd = dup(h);
  if (*fd < 0) { err = errno_str("dup"); return 1; }
  return 0;
}
inline int import_fd(int fd, Handle *h, std::string &err) {
  *h = dup(fd);
  if (*h < 0) { err = errno_str("dup"); return 1; }
  return 0;
}
inline int map(int, uint8_t *va, uint64_t bytes, Handle h, std::string &err) {
  if (mmap(va, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, h, 0) == MAP_FAILED) { err = errno_str("mmap (map)"); return 1; }
  return 0;
}
inline int unmap(uint8_t *va, uint64_t bytes, std::string &err) {
  if (mmap(va, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0) == MAP_FAILED) { err = errno_str("mmap (unmap)"); return 1; }
  return 0;
}
inline int release(Handle h, std::string &) { close(h); return 0; }
inline int unreserve(uint8_t *va, uint64_t bytes, std::string &err) {
  if (munmap(va, bytes)) { err = errno_str("munmap"); return 1; }
  return 0;
}
#endif

// ---- descriptors between the ranks of a node ---------------------------------------------------------------------------
// One connected Unix stream socket per pair of ranks.  A rank listens on an abstract name made from a random word; the
// words are exchanged through the codec's transport, after which rank r connects to every lower rank and accepts the
// higher ones.  Each pair's stream is ordered, so successive exchanges (attach, growths) cannot be confused.
struct FdMesh {
  uint32_t rank = 0, world = 1;
  int listener = -1;
  uint64_t word = 0;
  std::vector<int> peer;   // [world] connected socket to rank q (-1 for this rank)
};
// a rank that has died must make its peers fail, not wait forever: every socket operation of the mesh gives up after two minutes
inline void mesh_timeouts(int fd) {
  timeval tv;
  tv.tv_sec = 120; tv.tv_usec = 0;
  (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  (void)setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}
inline void mesh_name(uint64_t word, sockaddr_un *a, socklen_t *len) {
  memset(a, 0, sizeof(*a));
  a->sun_family = AF_UNIX;
  const int n = snprintf(a->sun_path + 1, sizeof(a->sun_path) - 1, "fqsx-%016llx", (unsigned long long)word);   // abstract: sun_path[0] = 0
  *len = (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + n);
}
inline int mesh_listen(FdMesh &m, uint32_t rank, uint32_t world, std::string &err) {
  m.rank = rank; m.world = world;
  m.peer.assign(world, -1);
  m.listener = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
  if (m.listener < 0) { err = errno_str("socket"); return 1; }
  int ur = open("/dev/urandom", O_RDONLY | O_CLOEXEC);
  for (int attempt = 0; attempt < 8; ++attempt) {
    uint64_t w = ((uint64_t)getpid() << 32) ^ (uint64_t)(uintptr_t)&m ^ ((uint64_t)attempt * 0x9E3779B97F4A7C15ull);
    if (ur >= 0 && read(ur, &w, sizeof(w)) != (ssize_t)sizeof(w)) w ^= (uint64_t)attempt;
    sockaddr_un a;
    socklen_t len;
    mesh_name(w, &a, &len);
    if (bind(m.listener, (sockaddr *)&a, len) == 0) { m.word = w; break; }
    if (errno != EADDRINUSE || attempt == 7) { err = errno_str("bind"); if (ur >= 0) close(ur); return 1; }
  }
  if (ur >= 0) close(ur);
  if (listen(m.listener, 256)) { err = errno_str("listen"); return 1; }
  mesh_timeouts(m.listener);   // (accept() honours SO_RCVTIMEO)
  return 0;
}
inline int write_all(int fd, const void *p, size_t n) {
  const char *c = (const char *)p;
  while (n) { ssize_t w = send(fd, c, n, MSG_NOSIGNAL); if (w <= 0) { if (w < 0 && errno == EINTR) continue; return 1; } c += w; n -= (size_t)w; }
  return 0;
}
inline int read_all(int fd, void *p, size_t n) {
  char *c = (char *)p;
  while (n) { ssize_t r = recv(fd, c, n, 0); if (r <= 0) { if (r < 0 && errno == EINTR) continue; return 1; } c += r; n -= (size_t)r; }
  return 0;
}
// words[q] = the word rank q listens on (all-gathered by the caller)
inline int mesh_connect(FdMesh &m, const uint64_t *words, std::string &err) {
  for (uint32_t q = 0; q < m.rank; ++q) {
    int s = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (s < 0) { err = errno_str("socket"); return 1; }
    sockaddr_un a;
    socklen_t len;
    mesh_name(words[q], &a, &len);
    mesh_timeouts(s);
    if (connect(s, (sockaddr *)&a, len)) { err = errno_str("connect to a rank of another node? (the partitioned mode is one node)"); close(s); return 1; }
    const uint32_t me = m.rank;
    if (write_all(s, &me, sizeof(me))) { err = errno_str("send"); close(s); return 1; }
    m.peer[q] = s;
  }
  for (uint32_t i = m.rank + 1; i < m.world; ++i) {
    int s = accept4(m.listener, nullptr, nullptr, SOCK_CLOEXEC);
    if (s < 0) { err = errno_str("accept (a rank of the world did not connect)"); return 1; }
    mesh_timeouts(s);
    uint32_t q = 0;
    if (read_all(s, &q, sizeof(q)) || q <= m.rank || q >= m.world || m.peer[q] >= 0) { err = "unexpected peer on the descriptor socket"; close(s); return 1; }
    m.peer[q] = s;
  }
  close(m.listener);
  m.listener = -1;
  return 0;
}
inline void mesh_close(FdMesh &m) {
  for (int &s : m.peer) if (s >= 0) { close(s); s = -1; }
  if (m.listener >= 0) { close(m.listener); m.listener = -1; }
}
#define FQSX_FD_BATCH 48
inline int send_fds(int sock, const int *fds, uint32_t n, std::string &err) {
  for (uint32_t o = 0; o < n; o += FQSX_FD_BATCH) {
    const uint32_t b = n - o < FQSX_FD_BATCH ? n - o : FQSX_FD_BATCH;
    char byte = 'F';
    iovec io = {&byte, 1};
    alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int) * FQSX_FD_BATCH)];
    memset(ctl, 0, sizeof(ctl));
    msghdr mh = {};
    mh.msg_iov = &io; mh.msg_iovlen = 1; mh.msg_control = ctl; mh.msg_controllen = CMSG_SPACE(sizeof(int) * b);
    cmsghdr *c = CMSG_FIRSTHDR(&mh);
    c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int) * b);
    memcpy(CMSG_DATA(c), fds + o, sizeof(int) * b);
    ssize_t w;
    do w = sendmsg(sock, &mh, MSG_NOSIGNAL); while (w < 0 && errno == EINTR);
    if (w != 1) { err = errno_str("sendmsg (descriptors)"); return 1; }
  }
  return 0;
}
inline int recv_fds(int sock, int *fds, uint32_t n, std::string &err) {
  for (uint32_t o = 0; o < n; o += FQSX_FD_BATCH) {
    const uint32_t b = n - o < FQSX_FD_BATCH ? n - o : FQSX_FD_BATCH;
    char byte = 0;
    iovec io = {&byte, 1};
    alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int) * FQSX_FD_BATCH)];
    msghdr mh = {};
    mh.msg_iov = &io; mh.msg_iovlen = 1; mh.msg_control = ctl; mh.msg_controllen = sizeof(ctl);
    ssize_t r;
    do r = recvmsg(sock, &mh, MSG_CMSG_CLOEXEC); while (r < 0 && errno == EINTR);
    if (r != 1) { err = errno_str("recvmsg (descriptors)"); return 1; }
    cmsghdr *c = CMSG_FIRSTHDR(&mh);
    if (!c || c->cmsg_level != SOL_SOCKET || c->cmsg_type != SCM_RIGHTS || c->cmsg_len != CMSG_LEN(sizeof(int) * b)) {
      err = "descriptor message of unexpected shape";
      return 1;
    }
    memcpy(fds + o, CMSG_DATA(c), sizeof(int) * b);
  }
  return 0;
}

}  // namespace fqsx_vm
