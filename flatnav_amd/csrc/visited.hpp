// visited.hpp -- part of the gfx950 search engine (device code; included only by beam_search.hip).
// Exact visited sets in LDS (32-bit open addressing and the 16-bit-tag bucketed table).
#pragma once
#include "search_params.h"
namespace fnv_dev {

// ---------------------------------------------------------------------------------------------
// Exact visited set: open addressing (linear probing) over uint32 ids in LDS, filled to at most
// 3/4; once it would exceed that, the remaining insertions of the query go to a per-slot HBM
// bitmap (one bit per node) that the slot clears again before its next query.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool visited_insert_lds(uint32_t* tab, uint32_t slots_mask, uint32_t shift, uint32_t id) {
  uint32_t h = (id * 0x9E3779B1u) >> shift;
  while (true) {
    uint32_t cur = tab[h];
    if (cur == id) return false;
    if (cur == EMPTY_ID) {
      uint32_t old = atomicCAS(&tab[h], EMPTY_ID, id);
      if (old == EMPTY_ID) return true;
      if (old == id) return false;
    }
    h = (h + 1) & slots_mask;
  }
}
__device__ __forceinline__ bool visited_lookup_lds(const uint32_t* tab, uint32_t slots_mask, uint32_t shift,
                                                   uint32_t id) {
  uint32_t h = (id * 0x9E3779B1u) >> shift;
  while (true) {
    uint32_t cur = tab[h];
    if (cur == id) return true;
    if (cur == EMPTY_ID) return false;
    h = (h + 1) & slots_mask;
  }
}

// Exact visited set in 16 bits per element (used whenever the id width allows it).
// ids < 2^nbits.  Two multiplicative hashes h_k(id) = (id * A_k) mod 2^nbits, A_k odd, are bijections
// on nbits-bit integers, so (bucket = top bits of h_k, rem = remaining low bits, k) identifies the
// id uniquely: the table stores only tag = ((rem << 1) | k) + 1 (0 = empty) -- no false positives.
// A bucket is four 16-bit tags (8 bytes); an id may sit in either of its two buckets (inserted into
// the emptier one).  If both buckets are full the id is recorded in the slot's HBM bitmap instead;
// buckets never lose entries, so "both full -> ask the bitmap" stays consistent for the whole query.
__device__ __forceinline__ bool has_tag(uint32_t w, uint32_t tag) {
  return (w & 0xFFFFu) == tag || (w >> 16) == tag;
}
__device__ __forceinline__ int zero_halves(uint32_t w) { return ((w & 0xFFFFu) == 0u) + ((w >> 16) == 0u); }

// Bucket geometry: buckets = mult * 2^k (mult 1 or 3, so tables of 2^j or 3*2^j slots exist), t = nbits - k.
// x = h * mult; bucket = x >> t; the low t bits of x, divided by mult, number the ids inside the bucket.
__device__ __forceinline__ void tag16_slot(const VisGeom& g, uint32_t h, uint32_t which, uint32_t& bucket,
                                           uint32_t& tag) {
  const uint32_t x = h * g.mult;
  bucket = x >> g.rshift;
  uint32_t rem = x & g.rmask;
  if (g.mult == 3) rem = (rem * 43691u) >> 17;  // rem / 3, exact below 2^16
  tag = (rem << 1) + 1u + which;
}

// The stash (search_params.h): lanes whose id found both its buckets full come here before the HBM bitmap (a wave-uniform
// branch around the call: rare).  A second, tiny table of FULL ids: STASH / 4 buckets of four (one 16-byte LDS read), the
// bucket chosen by a third hash; an id is looked up in its bucket, else put into the bucket's first free word (stored as
// id + 1, 0 = free), else -- bucket full -- left to the bitmap.  Nothing is ever removed, so "its stash bucket is full ->
// the bitmap decides" stays true for the rest of the query, exactly like "both table buckets full -> the stash decides":
// the three places stay consistent.  All lanes work at once (~30 instructions per call; a first version that compared one
// lane's id at a time against a fully associative stash cost ~50-100 per call and lost 10 % on 50M x 128 Gaussian rows,
// where queries sit for long with a nearly full stash).
__device__ __forceinline__ void visited_stash(uint32_t* ovf_list, uint32_t& to_bitmap, uint32_t id, uint32_t& isnew) {
#ifdef FNV_NO_STASH  // (experiment knob: every overflow straight to the bitmap, as in round 2)
  return;
#endif
  static_assert((STASH & (STASH - 1)) == 0 && STASH >= 8, "STASH / 4 buckets, a power of two");
  uint32_t* const bucket = ovf_list + OVF_LIST + 2 + 4u * ((id * 0xC2B2AE35u) >> (32 - (31 - __builtin_clz(STASH / 4))));
  const uint32_t key = id + 1u;
  uint32_t pend = to_bitmap;
  while (__ballot(pend != 0u) != 0ull) {
    const uint4 e = *reinterpret_cast<const uint4*>(bucket);
    const bool found = e.x == key || e.y == key || e.z == key || e.w == key;
    const int free_at = e.x == 0u ? 0 : e.y == 0u ? 1 : e.z == 0u ? 2 : e.w == 0u ? 3 : -1;
    if (pend && found) {
      to_bitmap = 0u;  // seen before
      pend = 0u;
    } else if (pend && free_at < 0) {
      pend = 0u;  // bucket full: the bitmap decides
    } else if (pend) {
      const uint32_t old = atomicCAS(bucket + free_at, 0u, key);
      if (old == 0u || old == key) {  // (old == key: another lane of this row holds the same id and was first)
        isnew |= old == 0u ? 1u : 0u;
        to_bitmap = 0u;
        pend = 0u;
      }  // else: another id took the word -- look again
    }
  }
}

// Called by ALL lanes (inactive ones pass act = false).  The probe is straight-line arithmetic (bitwise, no
// short-circuit branches) inside a wave-uniform retry loop that normally runs once, so EXEC is only touched
// around the CAS itself -- the scalar unit that manipulates EXEC is shared by every wave of the CU.
// bitmap / ovf_glist are this slot's HBM spill areas; they are only touched on the rare both-buckets-full path,
// where the list capacity is re-read from the kernel arguments (cold_args).
//
// Round 3 (instruction diet: the hop of 128-byte rows is bound by instruction issue, and this probe was ~130 of its
// ~560 instructions): a bucket fills in a fixed order -- low half of word 0, high half, low half of word 1, high half
// (an insertion takes the lowest empty half of the first word that has one, and nothing is ever removed) -- so
//   * "is the tag in one of the two buckets"  = some 16-bit half of {B1 ^ t1, B2 ^ t2} is zero: three packed 16-bit
//     minima over the four words, one more across the two halves (was: eight exact has-zero-halfword tests);
//   * "how full is a bucket" = position of its highest non-zero half, read off the leading zeros of the 64-bit word
//     (was: four has-zero tests and four popcounts);
//   * the slot to fill is that count: word = fill >> 1, half = fill & 1.
typedef unsigned short fnv_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  const fnv_u16x2 m = __builtin_elementwise_min(__builtin_bit_cast(fnv_u16x2, a), __builtin_bit_cast(fnv_u16x2, b));
  return __builtin_bit_cast(uint32_t, m);
}
// halves in use (0..4) of a bucket that fills from the low half of .x to the high half of .y
__device__ __forceinline__ uint32_t bucket_fill(const uint2& B) {
  const uint32_t top = B.y ? B.y : B.x;                  // the word that holds the highest used half
  return (B.y ? 2u : 0u) + (top ? 1u : 0u) + ((top >> 16) ? 1u : 0u);
}

__device__ __forceinline__ bool visited_insert_tag16(uint32_t* tab, const VisGeom& g, bool act, uint32_t id,
                                                     uint32_t* bitmap, uint32_t* ovf_list, uint32_t* ovf_glist,
                                                     bool& used_bitmap) {
  uint32_t b1, b2, t1, t2;
  tag16_slot(g, (id * 0x9E3779B1u) & g.nmask, 0u, b1, t1);
  tag16_slot(g, (id * 0x85EBCA6Bu) & g.nmask, 1u, b2, t2);
  const uint32_t t1x = t1 | (t1 << 16), t2x = t2 | (t2 << 16);  // the tag in both halves of a word
  uint32_t pending = act ? 1u : 0u, isnew = 0u;
  while (__ballot(pending != 0u) != 0ull) {
    const uint2 B1 = *reinterpret_cast<const uint2*>(tab + 2 * b1);
    const uint2 B2 = *reinterpret_cast<const uint2*>(tab + 2 * b2);
    // some half of the four xor-ed words is zero <=> the tag is there (tags are never zero, so an empty half is no hit)
    const uint32_t m = pk_min_u16(pk_min_u16(B1.x ^ t1x, B1.y ^ t1x), pk_min_u16(B2.x ^ t2x, B2.y ^ t2x));
    const uint32_t found = min(m & 0xFFFFu, m >> 16) == 0u ? 1u : 0u;
    const uint32_t f1 = bucket_fill(B1), f2 = bucket_fill(B2);
    const uint32_t full = min(f1, f2) == 4u ? 1u : 0u;
    const bool first = f1 <= f2;  // insert into the emptier bucket
    const uint32_t f = first ? f1 : f2;
    const uint32_t Bx = first ? B1.x : B2.x, By = first ? B1.y : B2.y;
    const uint32_t tag = first ? t1 : t2;
    const uint32_t oldw = (f & 2u) ? By : Bx;
    const uint32_t neww = oldw | (tag << ((f & 1u) * 16u));
    const uint32_t try_cas = pending & (found ^ 1u) & (full ^ 1u);
    uint32_t got = ~oldw;
    if (try_cas) got = atomicCAS(tab + 2 * (first ? b1 : b2) + (f >> 1), oldw, neww);
    const uint32_t won = try_cas & (got == oldw ? 1u : 0u);
    isnew |= won;
    uint32_t to_bitmap = pending & (found ^ 1u) & full;  // both buckets full: the stash, then the HBM bitmap decides (rare)
    if (__ballot(to_bitmap != 0u) != 0ull) visited_stash(ovf_list, to_bitmap, id, isnew);
#ifdef FNV_EXP_NO_BITMAP  // TIMING EXPERIMENT ONLY (wrong results): ids that overflow the table and the stash count as new,
    isnew |= to_bitmap;   // nothing goes to HBM -- what do the bitmap's round trips and DRAM operations cost a launch?
    to_bitmap = 0u;
#endif
    if (__ballot(to_bitmap != 0u) != 0ull) {
      used_bitmap = true;  // wave-uniform: set for every lane as soon as any lane's id goes to the bitmap
      if (to_bitmap) {
        const uint32_t bit = 1u << (id & 31);
        const uint32_t old = atomicOr(&bitmap[id >> 5], bit);
        if (!(old & bit)) {
          const uint32_t pos = atomicAdd(&ovf_list[0], 1u);
          // the first OVF_LIST ids are remembered so that a lightly used bitmap is cleared word by word; past
          // that the whole bitmap is cleared with wide sequential stores -- measured faster than scattered
          // 4-byte clears while the bitmap is small (125 KB at 1M nodes); for big indexes (ovf_cap > 0: bitmap
          // > 512 KB) a longer list in HBM keeps the clean-up proportional to the ids, not to N
          if (pos < OVF_LIST) ovf_list[1 + pos] = id;
          else if (pos - OVF_LIST < cold_args()->ovf_cap) ovf_glist[pos - OVF_LIST] = id;
          isnew = 1u;
        }
      }
    }
    pending = try_cas & (won ^ 1u);  // lost a race for that word: look again
  }
  return isnew != 0u;
}

// The same scheme with 64-bit buckets of three 21-bit or two 32-bit tags, for id ranges whose remainder does not fit
// 16 bits at an affordable bucket count (N > 2^24 at 4096 slots): tag width w needs nbits - k <= w - 2.
// Round 3: rewritten on the two 32-bit halves of the bucket word (it was 64-bit has-zero-field arithmetic with a 64-bit
// multiply to replicate the tag: ~150 instructions per link row; now ~70).  As in the 16-bit table a bucket fills from its
// lowest field upwards and nothing is ever removed, so "how full" is the number of non-zero fields and the slot to fill is
// that count; "is the tag there" is a three-way minimum of the xor-ed fields.
//   w = 21: fields at bits 0-20, 21-41, 42-62 of {hi, lo};   w = 32: the fields ARE lo and hi.
__device__ __forceinline__ bool visited_insert_tagw(unsigned long long* tab, const VisGeom& g, bool act, uint32_t id,
                                                    uint32_t* bitmap, uint32_t* ovf_list, uint32_t* ovf_glist,
                                                    bool& used_bitmap) {
  const uint32_t h1 = (id * 0x9E3779B1u) & g.nmask, h2 = (id * 0x85EBCA6Bu) & g.nmask;
  const uint32_t b1 = h1 >> g.rshift, b2 = h2 >> g.rshift;
  const uint32_t t1 = ((h1 & g.rmask) << 1) + 1u, t2 = ((h2 & g.rmask) << 1) + 2u;  // never zero; < 2^w
  const bool w21 = g.w == 21;  // wave-uniform
  constexpr uint32_t M21 = 0x1FFFFFu;
  uint2* tab2 = reinterpret_cast<uint2*>(tab);
  uint32_t pending = act ? 1u : 0u, isnew = 0u;
  while (__ballot(pending != 0u) != 0ull) {
    const uint2 B1 = tab2[b1], B2 = tab2[b2];
    uint32_t found, f1, f2;
    if (w21) {
      const uint32_t a0 = B1.x & M21, a1 = __builtin_amdgcn_alignbit(B1.y, B1.x, 21) & M21, a2 = (B1.y >> 10) & M21;
      const uint32_t c0 = B2.x & M21, c1 = __builtin_amdgcn_alignbit(B2.y, B2.x, 21) & M21, c2 = (B2.y >> 10) & M21;
      found = min(min(min(a0 ^ t1, a1 ^ t1), a2 ^ t1), min(min(c0 ^ t2, c1 ^ t2), c2 ^ t2)) == 0u ? 1u : 0u;
      f1 = min(a0, 1u) + min(a1, 1u) + min(a2, 1u);
      f2 = min(c0, 1u) + min(c1, 1u) + min(c2, 1u);
    } else {
      found = min(min(B1.x ^ t1, B1.y ^ t1), min(B2.x ^ t2, B2.y ^ t2)) == 0u ? 1u : 0u;
      f1 = min(B1.x, 1u) + min(B1.y, 1u);
      f2 = min(B2.x, 1u) + min(B2.y, 1u);
    }
    const uint32_t cap = w21 ? 3u : 2u;
    const uint32_t full = min(f1, f2) == cap ? 1u : 0u;
    const bool first = f1 <= f2;  // insert into the emptier bucket
    const uint32_t f = first ? f1 : f2, tag = first ? t1 : t2;
    const uint2 oldw = first ? B1 : B2;
    uint2 neww = oldw;
    if (w21) {  // field f of {hi, lo}: shift the tag left by 21 f
      neww.x |= f == 0u ? tag : f == 1u ? tag << 21 : 0u;
      neww.y |= f == 1u ? tag >> 11 : f == 2u ? tag << 10 : 0u;
    } else {
      neww.x |= f == 0u ? tag : 0u;
      neww.y |= f == 1u ? tag : 0u;
    }
    const unsigned long long old64 = ((unsigned long long)oldw.y << 32) | oldw.x;
    const unsigned long long new64 = ((unsigned long long)neww.y << 32) | neww.x;
    const uint32_t try_cas = pending & (found ^ 1u) & (full ^ 1u);
    unsigned long long got = ~old64;
    if (try_cas) got = atomicCAS(tab + (first ? b1 : b2), old64, new64);
    const uint32_t won = try_cas & (got == old64 ? 1u : 0u);
    isnew |= won;
    uint32_t to_bitmap = pending & (found ^ 1u) & full;
    if (__ballot(to_bitmap != 0u) != 0ull) visited_stash(ovf_list, to_bitmap, id, isnew);
#ifdef FNV_EXP_NO_BITMAP  // TIMING EXPERIMENT ONLY (wrong results): ids that overflow the table and the stash count as new,
    isnew |= to_bitmap;   // nothing goes to HBM -- what do the bitmap's round trips and DRAM operations cost a launch?
    to_bitmap = 0u;
#endif
    if (__ballot(to_bitmap != 0u) != 0ull) {
      used_bitmap = true;  // wave-uniform: set for every lane as soon as any lane's id goes to the bitmap
      if (to_bitmap) {
        const uint32_t bit = 1u << (id & 31);
        const uint32_t old = atomicOr(&bitmap[id >> 5], bit);
        if (!(old & bit)) {
          const uint32_t pos = atomicAdd(&ovf_list[0], 1u);
          if (pos < OVF_LIST) ovf_list[1 + pos] = id;
          else if (pos - OVF_LIST < cold_args()->ovf_cap) ovf_glist[pos - OVF_LIST] = id;
          isnew = 1u;
        }
      }
    }
    pending = try_cas & (won ^ 1u);
  }
  return isnew != 0u;
}

// Round 5, small launches on small indexes (search_params.h, vis_w == 1): the visited set is a plain bitmap of ALL node ids in
// LDS -- one atomic OR with return per id: a single LDS round trip, nothing to overflow, no stash, no HBM bitmap.  (A lone wave
// spends 1.8 k of its hop's 7.9 k cycles in the tag table: profiles/r5_phase_cycles_c2.txt.)
__device__ __forceinline__ bool visited_insert_direct(uint32_t* tab, bool act, uint32_t id) {
  uint32_t old = ~0u;
  if (act) old = atomicOr(tab + (id >> 5), 1u << (id & 31u));
  return ((old >> (id & 31u)) & 1u) == 0u;
}

// The visited set of the slot.  DIRECT is a property of the KERNEL (its own instantiations, launched for small launches only):
// a run-time third form next to the two tag tables cost the loaded hop of the bench shapes 1-4 % (code layout; r5_run32).
template <bool DIRECT>
__device__ __forceinline__ bool visited_insert(uint32_t* tab, const VisGeom& g, bool act, uint32_t id, uint32_t* bitmap,
                                               uint32_t* ovf_list, uint32_t* ovf_glist, bool& used_bitmap) {
  if constexpr (DIRECT) {
    return visited_insert_direct(tab, act, id);
  } else {
    if (g.w == 16) return visited_insert_tag16(tab, g, act, id, bitmap, ovf_list, ovf_glist, used_bitmap);
    return visited_insert_tagw(reinterpret_cast<unsigned long long*>(tab), g, act, id, bitmap, ovf_list, ovf_glist, used_bitmap);
  }
}

}  // namespace fnv_dev
