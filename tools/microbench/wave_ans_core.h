// generated from kernels.hip (WaveBits / WaveChan / WaveSegment) + experimental variants
struct WaveBits {           // uniform state; `win`: this lane's word of the current 64-word window
  const uint32_t* words;
  uint32_t wend, wbase, widx;      // window = words [wbase, wbase + 64); widx: the next word to take, relative to wbase
  uint32_t win;                    // (no second window in flight: a load pending across the sample loop puts an s_waitcnt vmcnt(0) — which also waits for the row stores — into every
                                   // iteration; the switch waits for its own load instead, ~1 us per 2048 bits)
  uint64_t buf;
  int avail;
  __device__ __forceinline__ uint32_t Load(uint32_t i) const { return i < wend ? LdG(words + i) : 0u; }
  __device__ __forceinline__ void Start(const uint32_t* w, uint32_t wend_, uint64_t bit_pos, uint32_t lane) {
    words = w; wend = wend_; wbase = (uint32_t)(bit_pos >> 5); widx = 0;
    win = Load(wbase + lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    buf = 0; avail = 0;
    Refill();
    const int skip = (int)(bit_pos & 31);
    buf >>= skip; avail -= skip;
    Refill();
  }
  __device__ __forceinline__ void Refill() {
    if (avail <= 32) {
      if (__builtin_expect(widx == 64, 0)) { const uint32_t lane = threadIdx.x & 63; wbase += 64; widx = 0; win = Load(wbase + lane); __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) here, so that none is needed in the loop */ }
      const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)widx);
      buf |= (uint64_t)w << avail;
      avail += 32;
      widx++;
    }
  }
  __device__ __forceinline__ uint64_t BitPos() const { return (uint64_t)(wbase + widx) * 32 - (uint64_t)avail; }
};
struct WaveChan {           // per channel (uniform unless noted)
  int32_t thr;              // per lane: split constant of this lane's inner node (INT_MAX beyond the last)
  uint32_t abase, cbase;    // per lane: LDS byte offsets of the wide table / cutoff table of this lane's interval's cluster
  uint32_t cluster;         // per lane: that cluster
  uint32_t la, cfg_off, cfg_uniform;
};
template <bool NEEDN, bool PROP9, int UPRED>
__device__ __forceinline__ void WaveSegment(WaveBits& bits, uint32_t& state, int32_t& left, int32_t& nw, const int32_t prevv, int32_t& curv, const int n, const WaveChan& wc) {
  const uint32_t la = wc.la, pmask = (1u << (12 - la)) - 1, lane = threadIdx.x & 63;
  int32_t cur = curv;
  for (int xl = 0; xl < n; xl++) {
    const int32_t W = left;
    int32_t N = W, NW = W;
    if (NEEDN) { N = __builtin_amdgcn_readlane(prevv, xl); NW = nw; nw = N; }
    const int32_t v0 = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
    int k = 0;
    if (PROP9) k = __builtin_popcountll(__ballot(v0 > wc.thr));
    int32_t guess;
    if (UPRED == 0) guess = 0;
    else if (UPRED == 1) guess = W;
    else { const int32_t m = min(N, W), M = max(N, W); guess = max(m, min(M, v0)); }   // clamped gradient = median(N, W, N + W - NW)
    // --- ANS: every lane reads the slot of its own interval's cluster
    const uint32_t slot = (state & 0xFFF) >> (12 - la), pos = state & pmask, hi = state >> 12;
    const uint2 e = LdS<uint2>(wc.abase + slot * 8);
    const uint32_t cr = LdS<uint16_t>(wc.cbase + slot * 2);
    const bool hit = pos >= (cr & 0xFFu);
    const uint32_t cand = hit ? e.y : e.x;
    const uint32_t sw = (uint32_t)__builtin_amdgcn_readlane((int)cand, k);
    state = (sw & 0xFFFu) * hi + (hi + pos) + ((sw >> 12) & 0xFFFu);
    int32_t v = (int32_t)sw >> 24;
    if (state < (1u << 16)) { asm volatile("" ::: "memory"); /* (keeps this a branch: as selects it costs 13 scalar instructions on every sample) */ state = (state << 16) | (uint32_t)(bits.buf & 0xFFFFu); bits.buf >>= 16; bits.avail -= 16; }
    if (__builtin_expect(v == kWideEscape, 0)) {
      // the token carries extra bits (or is too large for the table's byte): the symbol again from this lane's {cutoff, aliased symbol}, then dec_ans.h's hybrid integer
      const uint32_t crk = (uint32_t)__builtin_amdgcn_readlane((int)cr, k);
      uint32_t tok = pos >= (crk & 0xFFu) ? (crk >> 8) : slot;
      uint32_t cfg = wc.cfg_uniform;
      if (cfg == 0xFFFFFFFFu) cfg = Uniform(LdS<uint32_t>(wc.cfg_off + 4 * (uint32_t)__builtin_amdgcn_readlane((int)wc.cluster, k)));
      const uint32_t split_exp = cfg & 0xFF, split = 1u << split_exp;
      if (tok >= split) {
        const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
        const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb))) & 31;
        const uint32_t low = tok & ((1u << lsb) - 1);
        tok >>= lsb;
        if ((int)nbits > bits.avail) bits.Refill();
        const uint32_t xb = (uint32_t)(bits.buf & ((1ull << nbits) - 1));
        bits.buf >>= nbits; bits.avail -= (int)nbits;
        const uint32_t hb = (1u << msb) | (tok & ((1u << msb) - 1));
        tok = (((hb << nbits) | xb) << lsb) | low;
      }
      v = UnpackSigned(tok);
    }
    const int32_t val = (int32_t)((uint32_t)v + (uint32_t)guess);
    cur = (int)lane == xl ? val : cur;      // (v_writelane would need its value in a scalar register and its lane select in M0 on gfx9: three instructions against these two)
    left = val;
    bits.Refill();
  }
  curv = cur;
}

// ---- experimental variants --------------------------------------------------------------------------------------------------------------
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ int32_t WaveShr1(int32_t v, int32_t lane0) {     // lane i <- lane i - 1 of v; lane 0 <- lane0
  return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xF, 0xF, false);
}
// Variant 1: source written in the order the chain should issue, pinned with scheduling barriers: the alias reads go out first, the context of the
// sample (k, prediction) and the store of the sample before are computed in their shadow; refill only after bits were taken; N - NW of the whole
// segment from one DPP shift.
template <bool NEEDN, bool PROP9, int UPRED>
__device__ __forceinline__ void WaveSegment1(WaveBits& bits, uint32_t& state, int32_t& left, int32_t& nw, const int32_t prevv, int32_t& curv, const int n, const WaveChan& wc) {
  const uint32_t la = wc.la, pmask = (1u << (12 - la)) - 1, lane = threadIdx.x & 63;
  const uint32_t sh = 12 - la;
  int32_t cur = curv;
  int32_t dvec = 0;
  if (NEEDN) { dvec = prevv - WaveShr1(prevv, nw); nw = __builtin_amdgcn_readlane(prevv, 63); }
  int32_t val_prev = 0; int xl_prev = -1;
  for (int xl = 0; xl < n; xl++) {
    // [A] alias reads
    const uint32_t slot = (state & 0xFFF) >> sh;
    const uint2 e = LdS<uint2>(wc.abase + slot * 8);
    const uint32_t cr = LdS<uint16_t>(wc.cbase + slot * 2);
    SB();
    const uint32_t pos = state & pmask, hi = state >> 12, hp = hi + pos;
    SB();
    // [C] the sample before goes to its lane
    cur = (int)lane == xl_prev ? val_prev : cur;
    SB();
    // [B] context
    const int32_t W = left;
    int32_t N = W, v0 = W;
    if (NEEDN) { N = __builtin_amdgcn_readlane(prevv, xl); v0 = (int32_t)((uint32_t)W + (uint32_t)__builtin_amdgcn_readlane(dvec, xl)); }
    int k = 0;
    if (PROP9) k = __builtin_popcountll(__ballot(v0 > wc.thr));
    int32_t guess;
    if (UPRED == 0) guess = 0;
    else if (UPRED == 1) guess = W;
    else { const int32_t m = min(N, W), M = max(N, W); guess = max(m, min(M, v0)); }
    SB();
    const bool hit = pos >= (cr & 0xFFu);
    const uint32_t cand = hit ? e.y : e.x;
    const uint32_t sw = (uint32_t)__builtin_amdgcn_readlane((int)cand, k);
    state = (sw & 0xFFFu) * hi + hp + ((sw >> 12) & 0xFFFu);
    int32_t v = (int32_t)sw >> 24;
    SB();
    if (state < (1u << 16)) { asm volatile("" ::: "memory"); state = (state << 16) | (uint32_t)(bits.buf & 0xFFFFu); bits.buf >>= 16; bits.avail -= 16; bits.Refill(); }
    if (__builtin_expect(v == kWideEscape, 0)) {
      const uint32_t crk = (uint32_t)__builtin_amdgcn_readlane((int)cr, k);
      uint32_t tok = pos >= (crk & 0xFFu) ? (crk >> 8) : slot;
      uint32_t cfg = wc.cfg_uniform;
      if (cfg == 0xFFFFFFFFu) cfg = Uniform(LdS<uint32_t>(wc.cfg_off + 4 * (uint32_t)__builtin_amdgcn_readlane((int)wc.cluster, k)));
      const uint32_t split_exp = cfg & 0xFF, split = 1u << split_exp;
      if (tok >= split) {
        const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
        const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb))) & 31;
        const uint32_t low = tok & ((1u << lsb) - 1);
        tok >>= lsb;
        if ((int)nbits > bits.avail) bits.Refill();
        const uint32_t xb = (uint32_t)(bits.buf & ((1ull << nbits) - 1));
        bits.buf >>= nbits; bits.avail -= (int)nbits;
        const uint32_t hb = (1u << msb) | (tok & ((1u << msb) - 1));
        tok = (((hb << nbits) | xb) << lsb) | low;
        bits.Refill();
      }
      v = UnpackSigned(tok);
    }
    const int32_t val = (int32_t)((uint32_t)v + (uint32_t)guess);
    left = val; val_prev = val; xl_prev = xl;
  }
  cur = (int)lane == xl_prev ? val_prev : cur;
  curv = cur;
}
// Variant 2 (= 1 + values pinned where they are computed): source written in the order the chain should issue, pinned with scheduling barriers: the alias reads go out first, the context of the
// sample (k, prediction) and the store of the sample before are computed in their shadow; refill only after bits were taken; N - NW of the whole
// segment from one DPP shift.
template <bool NEEDN, bool PROP9, int UPRED>
__device__ __forceinline__ void WaveSegment2(WaveBits& bits, uint32_t& state, int32_t& left, int32_t& nw, const int32_t prevv, int32_t& curv, const int n, const WaveChan& wc) {
  const uint32_t la = wc.la, pmask = (1u << (12 - la)) - 1, lane = threadIdx.x & 63;
  const uint32_t sh = 12 - la;
  int32_t cur = curv;
  int32_t dvec = 0;
  if (NEEDN) { dvec = prevv - WaveShr1(prevv, nw); nw = __builtin_amdgcn_readlane(prevv, 63); }
  int32_t val_prev = 0; int xl_prev = -1;
  for (int xl = 0; xl < n; xl++) {
    // [A] alias reads
    const uint32_t slot = (state & 0xFFF) >> sh;
    const uint2 e = LdS<uint2>(wc.abase + slot * 8);
    const uint32_t cr = LdS<uint16_t>(wc.cbase + slot * 2);
    SB();
    const uint32_t pos = state & pmask, hi = state >> 12, hp = hi + pos;
    SB();
    // [C] the sample before goes to its lane
    cur = (int)lane == xl_prev ? val_prev : cur;
    asm volatile("" : "+v"(cur));
    SB();
    // [B] context
    const int32_t W = left;
    int32_t N = W, v0 = W;
    if (NEEDN) { N = __builtin_amdgcn_readlane(prevv, xl); v0 = (int32_t)((uint32_t)W + (uint32_t)__builtin_amdgcn_readlane(dvec, xl)); }
    int k = 0;
    if (PROP9) k = __builtin_popcountll(__ballot(v0 > wc.thr));
    int32_t guess;
    if (UPRED == 0) guess = 0;
    else if (UPRED == 1) guess = W;
    else { const int32_t m = min(N, W), M = max(N, W); guess = max(m, min(M, v0)); }
    asm volatile("" : "+v"(guess), "+s"(k));
    SB();
    const bool hit = pos >= (cr & 0xFFu);
    const uint32_t cand = hit ? e.y : e.x;
    const uint32_t sw = (uint32_t)__builtin_amdgcn_readlane((int)cand, k);
    state = (sw & 0xFFFu) * hi + hp + ((sw >> 12) & 0xFFFu);
    int32_t v = (int32_t)sw >> 24;
    SB();
    if (state < (1u << 16)) { asm volatile("" ::: "memory"); state = (state << 16) | (uint32_t)(bits.buf & 0xFFFFu); bits.buf >>= 16; bits.avail -= 16; bits.Refill(); }
    if (__builtin_expect(v == kWideEscape, 0)) {
      const uint32_t crk = (uint32_t)__builtin_amdgcn_readlane((int)cr, k);
      uint32_t tok = pos >= (crk & 0xFFu) ? (crk >> 8) : slot;
      uint32_t cfg = wc.cfg_uniform;
      if (cfg == 0xFFFFFFFFu) cfg = Uniform(LdS<uint32_t>(wc.cfg_off + 4 * (uint32_t)__builtin_amdgcn_readlane((int)wc.cluster, k)));
      const uint32_t split_exp = cfg & 0xFF, split = 1u << split_exp;
      if (tok >= split) {
        const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
        const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb))) & 31;
        const uint32_t low = tok & ((1u << lsb) - 1);
        tok >>= lsb;
        if ((int)nbits > bits.avail) bits.Refill();
        const uint32_t xb = (uint32_t)(bits.buf & ((1ull << nbits) - 1));
        bits.buf >>= nbits; bits.avail -= (int)nbits;
        const uint32_t hb = (1u << msb) | (tok & ((1u << msb) - 1));
        tok = (((hb << nbits) | xb) << lsb) | low;
        bits.Refill();
      }
      v = UnpackSigned(tok);
    }
    const int32_t val = (int32_t)((uint32_t)v + (uint32_t)guess);
    left = val; val_prev = val; xl_prev = xl;
  }
  cur = (int)lane == xl_prev ? val_prev : cur;
  curv = cur;
}
template <int V, bool NEEDN, bool PROP9, int UPRED>
__device__ __forceinline__ void WaveSegmentV(WaveBits& bits, uint32_t& state, int32_t& left, int32_t& nw, const int32_t prevv, int32_t& curv, const int n, const WaveChan& wc) {
  if (V == 0) WaveSegment<NEEDN, PROP9, UPRED>(bits, state, left, nw, prevv, curv, n, wc);
  else if (V == 1) WaveSegment1<NEEDN, PROP9, UPRED>(bits, state, left, nw, prevv, curv, n, wc);
  else WaveSegment2<NEEDN, PROP9, UPRED>(bits, state, left, nw, prevv, curv, n, wc);
}
