// One-shot and two-shot all-reduce for the small messages of tensor parallelism on MI355X (gfx950).
//
// The reference sums the partial results of every row-parallel matmul over the tensor-parallel
// group with NCCL's all-reduce replayed from a CUDA graph
// (vllm/model_executor/parallel_utils/parallel_state.py:506-556; call site
// parallel_utils/tensor_parallel/layers.py:416-426).  Those messages are small — a decode step
// reduces [256, 4096] fp16 = 2 MiB twice per layer, a small batch a few KiB — and on an MI355X
// node every pair of GPUs has an xGMI link of its own, so a ring (2 (N - 1) dependent hops over
// one link at a time) is the wrong shape for them.  Here every rank PUSHES its buffer straight
// into an inbox on every peer, all N - 1 links at once, raises a flag there, waits for the N - 1
// flags raised at home, and sums the N contributions out of local memory:
//
//   one launch, one hop, no second barrier.
//
// Why one barrier is enough: the inboxes are double buffered by the parity of the reduce's
// sequence number.  A peer can only start pushing reduce k + 1 (other parity) while this rank
// still sums reduce k, never k + 2: reduce k + 1 does not finish anywhere before this rank has
// raised its k + 1 flags, which it does only after its k sums are done.  The result is written
// in place over the caller's buffer, which no peer ever touches.  Every rank adds the N
// contributions in rank order in fp32, so all ranks hold bit-identical sums.
//
// Memory: peers' inboxes and flag words are reached through IPC mappings (one process per GPU)
// or plain pointers (several communicators in one process).  Nothing here uses a release or
// acquire FENCE: on this chip a fence at agent or system scope writes back or invalidates a
// whole L2, which stalls every other kernel on the GPU (measured in the decode kernel: a few
// hundred workgroups fencing once each made a 170 us launch take 1.7 ms) — and the all-reduce
// is meant to run beside the attention kernels.  Instead every access to memory that another
// rank writes or reads bypasses the caches at system scope (16-byte `sc0 sc1` stores that write
// through to the destination, 16-byte `sc0 sc1` loads that pass the caches: one full-width
// packet per lane on the link, not two), a wave waits for its stores
// to be acknowledged (s_waitcnt vmcnt(0)) before its workgroup counts as arrived, and the
// flags are relaxed system-scope atomics in uncached memory.  Every wait is bounded (the
// communicator's wait budget, pa_comm_set_wait_budget, in ticks of the 100 MHz wall clock;
// kDefaultWaitTicks unless set): a rank whose peers do not arrive within it gives up, counts a
// timeout that pa_comm_status reports, and finishes the launch with a result of NaNs rather
// than hanging the GPU.  From then on the communicator is broken for good: later reduces on it
// do not wait (a peer whose flag is already there is still summed, every other reduce gives NaN
// and is counted), so the cost of a dead peer is one bounded wait, not one per reduce; recovery
// is a new communicator.  The budget must cover the longest host-side stall a healthy rank can
// have between two reduces (graph capture, code-object load, a profiler attaching): the callers
// check pa_comm_status where they synchronise anyway and stop on a timeout instead of serving NaNs.
//
// The sequence number lives in device memory, not in the kernel arguments, so a captured launch
// can be replayed from a HIP graph.
//
// Two-shot form, for messages big enough that link time outweighs a second barrier: rank r owns
// the r-th N-th of the message.  Every rank pushes each peer ITS chunk only, the owners sum
// their chunks (rank order, fp32) and push the sums back out, a second round of flags, and
// everybody copies the N - 1 foreign chunks home: 2 / N of the message per link instead of all
// of it.  Same inboxes (first half of a slot: the contribution, second half: the owner's sum),
// same parity argument, a second flag array.
#include <hip/hip_runtime.h>
#include <string.h>

#include <new>

#include "pa_common.h"

namespace pa {

constexpr int kCommMaxRanks = 8;
constexpr int kCommThreads = 256;
constexpr int kCommMaxWorkgroups = 128;
constexpr unsigned long long kTicksPerSecond = 100000000ull;           // the wall clock (s_memrealtime) runs at 100 MHz
constexpr unsigned long long kDefaultWaitTicks = 30 * kTicksPerSecond;  // serving default: 30 s
constexpr size_t kTwoShotSavedBytes = 256 * 1024;         // see pa_comm_all_reduce

// control words of a rank, in uncached device memory
enum : int {
  kFlag0 = 0,                     // [8] contributions delivered (both forms)
  kFlagB0 = kCommMaxRanks,        // [8] owners' sums delivered (two-shot form)
  kSeq = 2 * kCommMaxRanks, kArrived, kArrivedB, kFinished, kTimeouts,
  kCtrlWords = 32
};

struct CommArgs {
  uint32_t* ctrl[kCommMaxRanks];  // every rank's control words, as mapped here
  char* inbox[kCommMaxRanks];     // every rank's inbox [2][world][slot_bytes], as mapped here
  void* data;                     // this rank's buffer: contribution in, sum out
  size_t slot_bytes;
  size_t n16;                     // 16-byte vectors in the message
  unsigned long long wait_ticks;  // how long a wait for a peer may last
  int rank;
  int world;
};

template <typename T> struct Acc;
template <> struct Acc<float> {
  static constexpr int N = 4;
  static constexpr uint32_t kNaN = 0x7fc00000u;
  static __device__ __forceinline__ void add(float* acc, u32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] += __uint_as_float(v[i]);
  }
  static __device__ __forceinline__ u32x4 pack(const float* acc) {
    return u32x4{__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
  }
};
template <> struct Acc<_Float16> {
  static constexpr int N = 8;
  static constexpr uint32_t kNaN = 0x7e007e00u;
  static __device__ __forceinline__ void add(float* acc, u32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f16x2 h = __builtin_bit_cast(f16x2, (uint32_t)v[i]);
      acc[2 * i] += (float)h[0];
      acc[2 * i + 1] += (float)h[1];
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float* acc) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f16x2 h = {(_Float16)acc[2 * i], (_Float16)acc[2 * i + 1]};
      r[i] = __builtin_bit_cast(uint32_t, h);
    }
    return r;
  }
};
template <> struct Acc<__bf16> {
  static constexpr int N = 8;
  static constexpr uint32_t kNaN = 0x7fc07fc0u;
  static __device__ __forceinline__ void add(float* acc, u32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc[2 * i] += __uint_as_float((uint32_t)v[i] << 16);
      acc[2 * i + 1] += __uint_as_float((uint32_t)v[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float* acc) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bf16x2 h = {(__bf16)acc[2 * i], (__bf16)acc[2 * i + 1]};
      r[i] = __builtin_bit_cast(uint32_t, h);
    }
    return r;
  }
};

// 16 bytes to / from memory shared with other ranks: ONE system-scope access that bypasses the
// caches (see the note on fences above) — buffer instructions with the sc0 and sc1 bits, which
// exist in 16-byte width (the widest atomic the compiler offers is 8 bytes, and two of those
// would put two packets per lane on an xGMI link).  A Window is the buffer descriptor of one
// rank's share of an inbox (base + 32-bit byte offsets; built from wave-uniform values).
constexpr int kAuxSystemScope = 17;  // sc0 | sc1
typedef unsigned int v4u32 __attribute__((__vector_size__(16)));
struct Window {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ explicit Window(const char* base)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)0xffffffffu, 0x00020000)) {}
  __device__ __forceinline__ void store(uint32_t byte_off, u32x4 v) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, v), rsrc, (int)byte_off, 0, kAuxSystemScope);
  }
  __device__ __forceinline__ u32x4 load(uint32_t byte_off) const {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, kAuxSystemScope));
  }
};

// The N contributions to a vector in rank order — the peers' from this rank's inbox (slot p at
// `mine + p * slot_bytes`, byte `at` of the slot), this rank's own from `own` — summed in fp32.
template <typename T>
__device__ __forceinline__ u32x4 sum_in_rank_order(const CommArgs& a, const char* mine, uint32_t at, u32x4 own) {
  u32x4 v[kCommMaxRanks];
#pragma unroll
  for (int p = 0; p < kCommMaxRanks; p++) {  // all the loads first: they are in flight together
    if (p < a.world && p != a.rank) v[p] = Window(mine + (size_t)p * a.slot_bytes).load(at);
  }
  float acc[Acc<T>::N];
#pragma unroll
  for (int e = 0; e < Acc<T>::N; e++) acc[e] = 0.f;
#pragma unroll
  for (int p = 0; p < kCommMaxRanks; p++) {
    if (p < a.world) Acc<T>::add(acc, p == a.rank ? own : v[p]);
  }
  return Acc<T>::pack(acc);
}

// The workgroup counts itself as arrived (every wave has released its stores); the last one to
// arrive raises this rank's flag `flag0 + rank` = k in every peer's control words.
__device__ __forceinline__ void arrive_and_signal(const CommArgs& a, uint32_t* ctrl, int counter, int flag0, uint32_t k,
                                                  int* last_to_arrive) {
  const int tid = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have been acknowledged
  __syncthreads();
  if (tid == 0) {
    const uint32_t before = __hip_atomic_fetch_add(&ctrl[counter], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_to_arrive = before + 1u == gridDim.x;
  }
  __syncthreads();
  if (*last_to_arrive) {
    if (tid < a.world && tid != a.rank)
      __hip_atomic_store(&a.ctrl[tid][flag0 + a.rank], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid == 0) __hip_atomic_store(&ctrl[counter], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Waits until every peer's flag `flag0 + peer` here has reached k (flags only grow), or gives up.
// `broken`: a wait of an earlier launch was given up.  The communicator stays broken from then on
// and waits for nobody, so that a dead peer costs one bounded wait, not one per reduce.
__device__ __forceinline__ void wait_for_peers(const CommArgs& a, uint32_t* ctrl, int flag0, uint32_t k, bool broken,
                                               int* gave_up) {
  const int tid = threadIdx.x;
  if (tid < a.world && tid != a.rank) {
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(&ctrl[flag0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - k) < 0) {
      if (broken || wall_clock64() - t0 > a.wait_ticks) {
        __hip_atomic_fetch_add(&ctrl[kTimeouts], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *gave_up = 1;  // LDS; every writer stores the same value
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();  // what the peers delivered is read with cache-bypassing loads from here on
}

// The last workgroup to finish closes the reduce.
__device__ __forceinline__ void finish(uint32_t* ctrl, uint32_t k) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t before = __hip_atomic_fetch_add(&ctrl[kFinished], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1u == gridDim.x) {
      __hip_atomic_store(&ctrl[kFinished], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctrl[kSeq], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <typename T, bool TWO_SHOT>
__global__ void __launch_bounds__(kCommThreads) pa_comm_all_reduce_kernel(const CommArgs a) {
  __shared__ int last_to_arrive;
  __shared__ int gave_up;  // a wait of this workgroup ended without its peer
  const int tid = threadIdx.x;
  if (tid == 0) gave_up = 0;
  uint32_t* ctrl = a.ctrl[a.rank];
  // written by the last workgroup of the previous reduce (or zero): stable for this whole launch
  const uint32_t k = __hip_atomic_load(&ctrl[kSeq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const bool broken = __hip_atomic_load(&ctrl[kTimeouts], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  const size_t parity_off = (size_t)(k & 1u) * a.world * a.slot_bytes;
  u32x4* data = static_cast<u32x4*>(a.data);
  const size_t first = (size_t)blockIdx.x * kCommThreads + tid;
  const size_t stride = (size_t)gridDim.x * kCommThreads;
  const char* mine = a.inbox[a.rank] + parity_off;
  const size_t my_slot = parity_off + (size_t)a.rank * a.slot_bytes;  // where this rank writes in a peer's inbox

  if (!TWO_SHOT) {
    // 1. push this rank's contribution into slot `rank` of every peer's inbox
    for (size_t i = first; i < a.n16; i += stride) {
      const u32x4 v = data[i];
#pragma unroll
      for (int p = 0; p < kCommMaxRanks; p++) {
        if (p < a.world && p != a.rank) Window(a.inbox[p] + my_slot).store((uint32_t)(i * 16), v);
      }
    }
    arrive_and_signal(a, ctrl, kArrived, kFlag0, k, &last_to_arrive);
    // 2. wait for every peer's contribution to this reduce
    wait_for_peers(a, ctrl, kFlag0, k, broken, &gave_up);
    // 3. sum the contributions in rank order, in place over this rank's own
    for (size_t i = first; i < a.n16; i += stride) data[i] = sum_in_rank_order<T>(a, mine, (uint32_t)(i * 16), data[i]);
  } else {
    // rank c owns vectors [c * q, (c + 1) * q) of the message
    const size_t q = (a.n16 + a.world - 1) / a.world;
    // 1. push every peer the part of this rank's contribution that the peer owns
    for (size_t i = first; i < a.n16; i += stride) {
      const int c = (int)(i / q);
      if (c != a.rank) Window(a.inbox[c] + my_slot).store((uint32_t)((i - (size_t)c * q) * 16), data[i]);
    }
    arrive_and_signal(a, ctrl, kArrived, kFlag0, k, &last_to_arrive);
    wait_for_peers(a, ctrl, kFlag0, k, broken, &gave_up);
    // 2. sum the chunk this rank owns, in rank order; keep it and push it to every peer
    const size_t lo = (size_t)a.rank * q, hi = lo + q < a.n16 ? lo + q : a.n16;
    for (size_t i = lo + first; i < hi; i += stride) {
      const u32x4 r = sum_in_rank_order<T>(a, mine, (uint32_t)((i - lo) * 16), data[i]);
      data[i] = r;
#pragma unroll
      for (int p = 0; p < kCommMaxRanks; p++) {
        if (p < a.world && p != a.rank) Window(a.inbox[p] + my_slot).store((uint32_t)((q + (i - lo)) * 16), r);
      }
    }
    arrive_and_signal(a, ctrl, kArrivedB, kFlagB0, k, &last_to_arrive);
    wait_for_peers(a, ctrl, kFlagB0, k, broken, &gave_up);
    // 3. bring the other owners' sums home
    for (size_t i = first; i < a.n16; i += stride) {
      const int c = (int)(i / q);
      if (c != a.rank) data[i] = Window(mine + (size_t)c * a.slot_bytes).load((uint32_t)((q + (i - (size_t)c * q)) * 16));
    }
  }
  // A reduce that did not hear from every peer must not pass for a sum: this workgroup's part of
  // the result becomes NaN (every workgroup waits for the same flags, so all of them get here).
  if (gave_up) {
    const u32x4 nan = {Acc<T>::kNaN, Acc<T>::kNaN, Acc<T>::kNaN, Acc<T>::kNaN};
    for (size_t i = first; i < a.n16; i += stride) data[i] = nan;
  }
  finish(ctrl, k);
}

}  // namespace pa

struct pa_comm {
  int rank = 0;
  int world = 1;
  int device = 0;
  size_t max_bytes = 0;
  size_t slot_bytes = 0;
  unsigned long long wait_ticks = pa::kDefaultWaitTicks;
  uint32_t* ctrl = nullptr;  // own control words
  char* inbox = nullptr;     // own inbox
  bool connected = false;
  bool peers_by_ipc = false;
  uint32_t* peer_ctrl[pa::kCommMaxRanks] = {};
  char* peer_inbox[pa::kCommMaxRanks] = {};
};

namespace pa {

#define PA_COMM_HIP(call, what)                                                           \
  do {                                                                                    \
    hipError_t pa_e_ = (call);                                                            \
    if (pa_e_ != hipSuccess) return fail(PA_ERR_HIP, "%s: %s", what, hipGetErrorString(pa_e_)); \
  } while (0)

static int comm_free(pa_comm* c) {
  if (c == nullptr) return PA_OK;
  if (c->peers_by_ipc) {
    for (int p = 0; p < c->world; p++) {
      if (p == c->rank) continue;
      if (c->peer_ctrl[p] != nullptr) (void)hipIpcCloseMemHandle(c->peer_ctrl[p]);
      if (c->peer_inbox[p] != nullptr) (void)hipIpcCloseMemHandle(c->peer_inbox[p]);
    }
  }
  if (c->ctrl != nullptr) (void)hipFree(c->ctrl);
  if (c->inbox != nullptr) (void)hipFree(c->inbox);
  delete c;
  return PA_OK;
}

}  // namespace pa

extern "C" int pa_comm_create(int rank, int world, size_t max_bytes, pa_comm** comm, unsigned char* handles_out) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(comm != nullptr && handles_out != nullptr, "pa_comm_create: null pointer argument");
  PA_REQUIRE(world >= 1 && world <= kCommMaxRanks && rank >= 0 && rank < world,
             "pa_comm_create: rank %d of %d (at most %d ranks)", rank, world, kCommMaxRanks);
  PA_REQUIRE(max_bytes > 0, "pa_comm_create: max_bytes must be positive");
  pa_comm* c = new (std::nothrow) pa_comm();
  if (c == nullptr) return fail(PA_ERR_HIP, "pa_comm_create: out of host memory");
  c->rank = rank;
  c->world = world;
  c->max_bytes = max_bytes;
  c->slot_bytes = (max_bytes + 255) / 256 * 256 + 256;  // the two-shot form rounds its chunks up
  memset(handles_out, 0, PA_COMM_HANDLE_BYTES);
  hipError_t e = hipGetDevice(&c->device);
  if (e == hipSuccess) e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->ctrl), 256, hipDeviceMallocUncached);
  if (e == hipSuccess) e = hipMemset(c->ctrl, 0, 256);
  // The inbox is written by other devices while kernels of this one read it: fine-grained memory
  // is what the runtime keeps coherent between devices during a kernel (the accesses themselves
  // are system-scope atomics either way).
  if (e == hipSuccess)
    e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->inbox), 2 * (size_t)world * c->slot_bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    const size_t inbox_bytes = 2 * (size_t)world * c->slot_bytes;
    comm_free(c);
    return fail(PA_ERR_HIP, "pa_comm_create: allocating the control words and a %zu-byte inbox: %s", inbox_bytes,
                hipGetErrorString(e));
  }
  // Handles for peers in other processes.  A communicator whose peers live in this process
  // (pa_comm_connect_local) never needs them: failing to export is not an error here.
  static_assert(2 * sizeof(hipIpcMemHandle_t) <= PA_COMM_HANDLE_BYTES, "handle blob too small");
  hipIpcMemHandle_t h[2];
  if (hipIpcGetMemHandle(&h[0], c->ctrl) == hipSuccess && hipIpcGetMemHandle(&h[1], c->inbox) == hipSuccess) {
    memcpy(handles_out, h, sizeof(h));
  } else {
    (void)hipGetLastError();
  }
  *comm = c;
  return PA_OK;
}

extern "C" int pa_comm_connect(pa_comm* c, const unsigned char* all_handles) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(c != nullptr && all_handles != nullptr, "pa_comm_connect: null pointer argument");
  PA_REQUIRE(!c->connected, "pa_comm_connect: already connected");
  for (int p = 0; p < c->world; p++) {
    if (p == c->rank) {
      c->peer_ctrl[p] = c->ctrl;
      c->peer_inbox[p] = c->inbox;
      continue;
    }
    hipIpcMemHandle_t h[2];
    memcpy(h, all_handles + (size_t)p * PA_COMM_HANDLE_BYTES, sizeof(h));
    bool blank = true;
    for (size_t i = 0; i < sizeof(h); i++) blank = blank && reinterpret_cast<const unsigned char*>(h)[i] == 0;
    PA_REQUIRE(!blank, "pa_comm_connect: rank %d exported no handle (hipIpcGetMemHandle failed there)", p);
    c->peers_by_ipc = true;
    PA_COMM_HIP(hipIpcOpenMemHandle(reinterpret_cast<void**>(&c->peer_ctrl[p]), h[0], hipIpcMemLazyEnablePeerAccess),
                "pa_comm_connect: hipIpcOpenMemHandle(control words)");
    PA_COMM_HIP(hipIpcOpenMemHandle(reinterpret_cast<void**>(&c->peer_inbox[p]), h[1], hipIpcMemLazyEnablePeerAccess),
                "pa_comm_connect: hipIpcOpenMemHandle(inbox)");
  }
  c->connected = true;
  return PA_OK;
}

extern "C" int pa_comm_connect_local(pa_comm* c, pa_comm* const* all) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(c != nullptr && all != nullptr, "pa_comm_connect_local: null pointer argument");
  PA_REQUIRE(!c->connected, "pa_comm_connect_local: already connected");
  for (int p = 0; p < c->world; p++) {
    PA_REQUIRE(all[p] != nullptr && all[p]->rank == p && all[p]->world == c->world && all[p]->slot_bytes == c->slot_bytes,
               "pa_comm_connect_local: entry %d is not rank %d of a communicator of the same shape", p, p);
    if (all[p]->device != c->device) {
      int can = 0;
      PA_COMM_HIP(hipDeviceCanAccessPeer(&can, c->device, all[p]->device), "pa_comm_connect_local: hipDeviceCanAccessPeer");
      PA_REQUIRE(can, "pa_comm_connect_local: device %d cannot access device %d", c->device, all[p]->device);
      hipError_t e = hipDeviceEnablePeerAccess(all[p]->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return fail(PA_ERR_HIP, "pa_comm_connect_local: hipDeviceEnablePeerAccess: %s", hipGetErrorString(e));
      (void)hipGetLastError();
    }
    c->peer_ctrl[p] = all[p]->ctrl;
    c->peer_inbox[p] = all[p]->inbox;
  }
  c->connected = true;
  return PA_OK;
}

extern "C" int pa_comm_all_reduce(pa_comm* c, void* data, size_t count, int dtype, int algorithm, void* stream) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(c != nullptr, "pa_comm_all_reduce: null communicator");
  if (dtype != PA_DTYPE_F32 && dtype != PA_DTYPE_F16 && dtype != PA_DTYPE_BF16)
    return fail(PA_ERR_UNSUPPORTED, "pa_comm_all_reduce: unsupported data type code: %d", dtype);
  if (count == 0 || c->world == 1) return PA_OK;
  PA_REQUIRE(c->connected, "pa_comm_all_reduce: communicator is not connected");
  PA_REQUIRE(data != nullptr && aligned16(data), "pa_comm_all_reduce: data must be a 16-byte aligned device pointer");
  const size_t bytes = count * (size_t)elem_size(dtype);
  PA_REQUIRE(bytes % 16 == 0, "pa_comm_all_reduce: message of %zu bytes is not a multiple of 16", bytes);
  PA_REQUIRE(bytes <= c->max_bytes, "pa_comm_all_reduce: message of %zu bytes exceeds the communicator's %zu", bytes,
             c->max_bytes);
  PA_REQUIRE(c->slot_bytes < (1ull << 31), "pa_comm_all_reduce: inbox slots of %zu bytes (offsets inside a slot are 32-bit)",
             c->slot_bytes);
  CommArgs a;
  for (int p = 0; p < kCommMaxRanks; p++) {
    a.ctrl[p] = p < c->world ? c->peer_ctrl[p] : nullptr;
    a.inbox[p] = p < c->world ? c->peer_inbox[p] : nullptr;
  }
  a.data = data;
  a.slot_bytes = c->slot_bytes;
  a.n16 = bytes / 16;
  a.wait_ticks = c->wait_ticks;
  a.rank = c->rank;
  a.world = c->world;
  PA_REQUIRE(algorithm == PA_COMM_AUTO || algorithm == PA_COMM_ONE_SHOT || algorithm == PA_COMM_TWO_SHOT,
             "pa_comm_all_reduce: unknown algorithm code %d", algorithm);
  // Two-shot when the link time it saves (a fraction (N - 2) / N of the message) outweighs its
  // second round of flags: from kTwoShotSavedBytes saved per link on.  Every rank decides alike.
  bool two_shot = algorithm == PA_COMM_TWO_SHOT;
  if (algorithm == PA_COMM_AUTO) two_shot = c->world > 2 && bytes / c->world * (c->world - 2) >= kTwoShotSavedBytes;
  if (c->world == 2 && algorithm == PA_COMM_AUTO) two_shot = false;
  // one vector per thread and pass; few workgroups (all resident at once, whatever else runs)
  size_t wgs = (a.n16 + kCommThreads - 1) / kCommThreads;
  if (wgs > (size_t)kCommMaxWorkgroups) wgs = kCommMaxWorkgroups;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define PA_COMM_LAUNCH(T)                                                                                          \
  do {                                                                                                             \
    if (two_shot)                                                                                                  \
      hipLaunchKernelGGL((pa_comm_all_reduce_kernel<T, true>), dim3((unsigned)wgs), dim3(kCommThreads), 0, st, a);  \
    else                                                                                                           \
      hipLaunchKernelGGL((pa_comm_all_reduce_kernel<T, false>), dim3((unsigned)wgs), dim3(kCommThreads), 0, st, a); \
  } while (0)
  switch (dtype) {
    case PA_DTYPE_F32: PA_COMM_LAUNCH(float); break;
    case PA_DTYPE_F16: PA_COMM_LAUNCH(_Float16); break;
    default: PA_COMM_LAUNCH(__bf16); break;
  }
#undef PA_COMM_LAUNCH
  PA_CHECK_LAUNCH("pa_comm_all_reduce_kernel launch");
  return PA_OK;
}

extern "C" int pa_comm_set_wait_budget(pa_comm* c, double seconds) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(c != nullptr, "pa_comm_set_wait_budget: null communicator");
  PA_REQUIRE(seconds > 0.0 && seconds <= 86400.0, "pa_comm_set_wait_budget: %g seconds (0 < budget <= 86400)", seconds);
  const double ticks = seconds * (double)kTicksPerSecond;
  c->wait_ticks = ticks < 1.0 ? 1ull : (unsigned long long)ticks;
  return PA_OK;
}

extern "C" int pa_comm_status(pa_comm* c, uint32_t* reduces_done, uint32_t* timeouts) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(c != nullptr, "pa_comm_status: null communicator");
  uint32_t words[kCtrlWords];
  PA_COMM_HIP(hipMemcpy(words, c->ctrl, sizeof(words), hipMemcpyDeviceToHost), "pa_comm_status: hipMemcpy");
  if (reduces_done != nullptr) *reduces_done = words[kSeq];
  if (timeouts != nullptr) *timeouts = words[kTimeouts];
  return PA_OK;
}

extern "C" int pa_comm_destroy(pa_comm* c) {
  pa::clear_error();
  return pa::comm_free(c);
}
