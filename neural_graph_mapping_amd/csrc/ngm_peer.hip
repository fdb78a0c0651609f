// One-shot exchange of the 16 loss sums / counts between the ranks of one node: the path's only collective
// (rm.py has no distributed code -- the exchange exists because the per-loss means of rm.py:1803-1871 are
// taken over ALL fields of the iteration and the fields are sharded).  RCCL's all-reduce of 64 bytes is a host-side
// call between two captured graphs; this kernel sits INSIDE the captured iteration instead:
//   every rank owns a mailbox of [2 parities][world][16] 8-byte words in fine-grained device memory, mapped into every
//   other rank's address space (hipIpc); word = (float value, 32-bit sequence number) written with ONE 64-bit store, so
//   that a reader never sees a value without its sequence number (the "LL" idea of the collective libraries);
//   rank r stores its 16 values into slot r of EVERY mailbox (its own included: 8 x 16 stores over xGMI, one per lane),
//   then polls its own mailbox until all `world` slots carry the current sequence number and sums them in rank order --
//   the same order on every rank, so all ranks hold bit-identical sums.
// Two parities: a rank can be at most one exchange ahead of the slowest (it cannot finish exchange k + 1 before every
// rank has written k + 1, i.e. read all of k), so exchange k + 2 never overwrites unread data of exchange k.
// The sequence number lives on the device and advances with every launch: graph replays need no host update.
// The poll is bounded (an unbounded spin of a kernel whose peer died would wedge the GPU): `max_spins` x ~1 us, 30 s by default
// (ngm_peer_set_timeout) -- long enough for a rank that writes a checkpoint, renders an evaluation image on rank 0 only or
// sits in a garbage collection, which the all-reduce this replaces would simply have waited for.  On expiry the kernel raises
// bit 0 of a STICKY status word and returns NaN in every sum: the losses of that iteration and the parameters its update
// touches become NaN on this rank -- visible in the same iteration, never a silently wrong normaliser (PeerExchange.check
// raises on the status; the renderer calls it before checkpoints and every few iterations).  A slot that already carries a LATER sequence
// number (possible only after such a time-out: the fast rank went on) is accepted and raises bit 1.  Once bit 0 is set the
// exchange is dead on this rank: every later launch still delivers its own words (the peers are not made to wait) but does not
// poll -- it returns NaN at once instead of waiting out another time-out per iteration until the host looks at the status
// (a dead peer used to cost peer_check_interval x time-out of wedged GPU before the RuntimeError).
#include "ngm_launch.h"

__global__ __launch_bounds__(128) void k_loss_exchange(ngm_peer_exchange px, float* sums, uint32_t max_spins) {
  const int t = threadIdx.x, slot = t & 15, peer = t >> 4;
  const uint32_t seq = (uint32_t)(*px.seq) + 1u;
  const int par = seq & 1u;
  const int W = px.world;
  if (peer < W) {
    const uint32_t bits = __float_as_uint(sums[slot]);
    unsigned long long* mb = reinterpret_cast<unsigned long long*>(px.mailbox[peer]);
    const unsigned long long word = ((unsigned long long)seq << 32) | bits;
    __hip_atomic_store(mb + ((size_t)par * NGM_MAX_PEERS + px.rank) * 16 + slot, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();          // this rank's `sums` are read before they are overwritten below
  if (t < 16) {
    const unsigned long long* mine = reinterpret_cast<const unsigned long long*>(px.mailbox[px.rank]) + (size_t)par * NGM_MAX_PEERS * 16;
    float total = 0.f;
    bool late = false, skew = false;
    const bool dead = (__hip_atomic_load(px.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) != 0;   // an earlier exchange timed out
    for (int p = 0; p < W && !dead; ++p) {
      unsigned long long w = 0;
      uint32_t spins = 0;
      for (;;) {
        w = __hip_atomic_load(mine + p * 16 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const int32_t ahead = (int32_t)((uint32_t)(w >> 32) - seq);
        if (ahead == 0) break;
        if (ahead > 0 && ahead < (1 << 30)) { skew = true; break; }   // the writer is past this exchange: resynchronise, flag it
        if (++spins > max_spins) { late = true; w = 0; break; }   // x ~1 us of s_sleep; the slot still holds the word of
                                                                            // two exchanges ago: contribute nothing, not that
        __builtin_amdgcn_s_sleep(32);
      }
      total += __uint_as_float((uint32_t)w);
    }
    if (late) atomicOr(px.status, 1);
    if (skew) atomicOr(px.status, 2);
    sums[t] = (late || dead) ? __uint_as_float(0x7fc00000u) : total;      // a time-out poisons the iteration instead of mis-normalising it
  }
  if (t == 0) *px.seq = (unsigned long long)seq;
}

static double g_peer_timeout_s = [] { const char* e = getenv("NGM_PEER_TIMEOUT_S"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 30.0; }();
double ngm_peer_set_timeout_impl(double seconds) {
  const double prev = g_peer_timeout_s;
  if (seconds > 0.0) g_peer_timeout_s = seconds;
  return prev;
}

int ngm_launch_loss_exchange(const ngm_peer_exchange& px, float* sums, hipStream_t st) {
  // one poll iteration = s_sleep(32) + a system-scope load: ~0.95 us measured ((1 << 21) spins = ~2 s in round 4's tests)
  const double spins = g_peer_timeout_s * (double)(1 << 20);
  const uint32_t max_spins = spins > 4.0e9 ? 4000000000u : (uint32_t)spins;
  hipLaunchKernelGGL(k_loss_exchange, dim3(1), dim3(128), 0, st, px, sums, max_spins);
  return 0;
}

// Mailbox memory: uncached / fine-grained device memory where the runtime offers it (stores of a RUNNING kernel on another
// GPU must become visible to the polling loads; coarse-grained memory only promises that at kernel boundaries), zeroed.
int ngm_peer_alloc_impl(int64_t bytes, void** out) {
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&p, (size_t)bytes); }
  if (e != hipSuccess) return (int)e;
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
  *out = p;
  return 0;
}
