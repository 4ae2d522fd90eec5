// Seed + probe of one read by one wavefront (SetSlotsVec + GetBlob for every k-mer of both strands, state1.cpp:95-127,
// ufindex.cpp GetBlob): shared by seed_probe_kernel (one launch for a batch) and search_pe_kernel (at the start of a pair).
#pragma once

#include "kernels.h"

#include "dev_common.h"

namespace urx {

// 1: the pair kernel's letter planes kept as ONE interleaved stream (probe_pair / slot_from_planes below); 0: two planes, interleaved per lane and k-mer (rounds 4-6)
#ifndef URX_SLOT_STREAM
#define URX_SLOT_STREAM 0  // measured: pairs 18.0 -> 18.3 ms per 1 M reads (the stream's words are made on the scalar unit at the start of every pair, the cuts it saves are rarer): off (profiles/r6/ab_instruction_trims.txt)
#endif

// The k-mers of NC 64-position chunks of one read: all slot numbers first, then all 2*NC slot loads in flight
// together, then the stores (the loads are random 64-byte sectors of a 26 GB table: their latency is the kernel).
template <int NCH, int NC, int C0 = 0>
__device__ __forceinline__ void probe_chunks(const DevIndex &X, const uint64_t (&lo)[NCH + 1], const uint64_t (&hi)[NCH + 1],
                                             const uint64_t (&inv)[NCH + 1], const uint64_t (&invm)[NCH + 1], int lane,
                                             uint32_t QL, uint32_t nwords, uint64_t base2, const ProbeOut &out) {
	const uint32_t W = X.W;
	uint64_t sp[NC], sm[NC];
	uint32_t rp[NC][2], rm[NC][2];
	bool vp[NC], vm[NC];
#pragma unroll
	for (int c = 0; c < NC; ++c)
		kmer_slots(X, lo[C0 + c], hi[C0 + c], inv[C0 + c], invm[C0 + c], lo[C0 + c + 1], hi[C0 + c + 1], inv[C0 + c + 1],
		           invm[C0 + c + 1], lane, 64u * (C0 + c) + lane, nwords, sp[c], sm[c], vp[c], vm[c]);
#pragma unroll
	for (int c = 0; c < NC; ++c) {  // lanes without a valid word fetch slot 0 (one cached sector) and drop it
		const uint64_t ap = vp[c] ? 5ull * sp[c] : 0ull, am = vm[c] ? 5ull * sm[c] : 0ull;
		const uint32_t *qp = reinterpret_cast<const uint32_t *>(X.blob + (ap & ~3ull));
		const uint32_t *qm = reinterpret_cast<const uint32_t *>(X.blob + (am & ~3ull));
		rp[c][0] = qp[0]; rp[c][1] = qp[1];
		rm[c][0] = qm[0]; rm[c][1] = qm[1];
	}
#pragma unroll
	for (int c = 0; c < NC; ++c) {
		const uint32_t p = 64u * (C0 + c) + lane;
		if (p >= nwords) continue;
		const uint64_t xp = (((uint64_t)rp[c][1] << 32) | rp[c][0]) >> (8u * (uint32_t)((5ull * sp[c]) & 3ull));
		const uint64_t xm = (((uint64_t)rm[c][1] << 32) | rm[c][0]) >> (8u * (uint32_t)((5ull * sm[c]) & 3ull));
		const uint64_t ip = base2 + p;
		const uint64_t im = base2 + QL + (QL - W - p);
		out.slots[ip] = vp[c] ? sp[c] : ~0ull;
		out.tallies[ip] = vp[c] ? (uint8_t)(xp & 0xFF) : (uint8_t)TALLY_FREE;
		out.positions[ip] = vp[c] ? (uint32_t)(xp >> 8) : 0xFFFFFFFFu;
		out.slots[im] = vm[c] ? sm[c] : ~0ull;
		out.tallies[im] = vm[c] ? (uint8_t)(xm & 0xFF) : (uint8_t)TALLY_FREE;
		out.positions[im] = vm[c] ? (uint32_t)(xm >> 8) : 0xFFFFFFFFu;
	}
}

// q = the read's bytes, off = its offset in the batch (the output index is 2 * off + strand * QL + qpos); QL in [W, 64 * NCH]
template <int NCH>
__device__ __forceinline__ void probe_read(const DevIndex &X, const uint8_t *__restrict__ q, uint32_t QL, uint64_t off, int lane,
                                           const ProbeOut &out) {
	const uint32_t W = X.W;

	// ballot planes: bit p of lo/hi = letter bits of base p, inv = base p is not ACGTU (or beyond the read)
	// invm: as inv for the reverse-complement strand -- lower-case 'u' complements to '?' (alpha.cpp:3005)
	uint64_t lo[NCH + 1], hi[NCH + 1], inv[NCH + 1], invm[NCH + 1];
	uint32_t chv[NCH];
#pragma unroll
	for (int c = 0; c < NCH; ++c) {  // every chunk's bytes are requested before the first is looked at
		const uint32_t p = 64u * c + lane;
		chv[c] = q[p < QL ? p : QL - 1];
	}
#pragma unroll
	for (int c = 0; c < NCH; ++c) {
		const uint32_t p = 64u * c + lane;
		const uint32_t ch = p < QL ? chv[c] : 0u;
		const uint32_t L = p < QL ? letter_of(ch) : 4u;
		lo[c] = __ballot(L & 1u);
		hi[c] = __ballot((L >> 1) & 1u);
		inv[c] = __ballot(L > 3u);
		invm[c] = __ballot(L > 3u || ch == 'u');
	}
	lo[NCH] = hi[NCH] = 0;
	inv[NCH] = invm[NCH] = ~0ull;

	const uint32_t nwords = QL - (W - 1);
	const uint64_t base2 = 2ull * off;
	const uint32_t nc = (nwords + 63u) >> 6;  // chunks that hold a k-mer start (same for the whole wave)
	if (nc <= 1) probe_chunks<NCH, 1>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	else if (NCH >= 2 && nc == 2) probe_chunks<NCH, (NCH >= 2 ? 2 : 1)>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	else if (NCH >= 3 && nc == 3) probe_chunks<NCH, (NCH >= 3 ? 3 : 1)>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	else if (NCH >= 4 && nc == 4) probe_chunks<NCH, (NCH >= 4 ? 4 : 1)>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	else if (NCH >= 5 && nc == 5) probe_chunks<NCH, (NCH >= 5 ? 5 : 1)>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	else if (NCH >= 6 && nc == 6) probe_chunks<NCH, (NCH >= 6 ? 6 : 1)>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	else if (NCH >= 7 && nc == 7) probe_chunks<NCH, (NCH >= 7 ? 7 : 1)>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	else if (NCH >= 8 && nc <= 8) probe_chunks<NCH, (NCH >= 8 ? 8 : 1)>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	else if constexpr (NCH >= 16) {  // long reads: two sweeps of eight chunks each
		probe_chunks<NCH, 8, 0>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
		probe_chunks<NCH, 8, 8>(X, lo, hi, inv, invm, lane, QL, nwords, base2, out);
	}
}

// Both mates of a pair by one wavefront, as search_pe_kernel wants them at the start of a pair: the bytes of the two mates
// are requested together, then all slot numbers are hashed, then every slot load of the pair is in flight at once (one
// memory round trip for the pair instead of one per mate), and the entries go straight into the kernel's LDS staging
// tables s_tal / s_pos ([mate][strand][qpos]; 0 / 0xFFFFFFFF beyond the last k-mer start).
// Round 4: nothing is written to HBM.  (Round 3 also stored slot / tally / position of every k-mer in the batch's probe
// arrays -- 13 B x 2 x 254 per pair, a third of the kernel's WRITE_SIZE -- for the pending stage and the rescue scan, which
// run long after the staging tables have been reused.)  What those stages need is the slot NUMBER of a few k-mers, and that
// is a function of the read: the letter planes of the two mates are kept in LDS (kpl[mate][plane][chunk]: low bit, high
// bit, not-a-letter, not-a-letter on the minus strand; 32 B per 64 bases) and slot_from_planes() re-derives a slot where it
// is needed; its tally and position are read from the table again (the first hop of the chain walk).
template <int NCH, int QMAX>
__device__ __forceinline__ void probe_pair(const DevIndex &X, const uint8_t *__restrict__ bases, const uint64_t (&off)[2], const uint32_t (&QL)[2],
                                           int lane, uint8_t (*__restrict__ s_tal)[2][QMAX], uint32_t (*__restrict__ s_pos)[2][QMAX],
                                           lds_ptr<uint64_t> kpl0, lds_ptr<uint64_t> kpl1) {
	const uint32_t W = X.W;
	uint32_t chv[2][NCH];
#pragma unroll
	for (int a = 0; a < 2; ++a)
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			const uint32_t p = 64u * c + lane;
			chv[a][c] = bases[off[a] + (p < QL[a] ? p : QL[a] - 1)];
		}
	uint64_t lo[2][NCH + 1], hi[2][NCH + 1], inv[2][NCH + 1], invm[2][NCH + 1];
#pragma unroll
	for (int a = 0; a < 2; ++a) {
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			const uint32_t p = 64u * c + lane;
			const uint32_t ch = p < QL[a] ? chv[a][c] : 0u;
			const uint32_t L = p < QL[a] ? letter_of(ch) : 4u;
			lo[a][c] = __ballot(L & 1u);
			hi[a][c] = __ballot((L >> 1) & 1u);
			inv[a][c] = __ballot(L > 3u);
			invm[a][c] = __ballot(L > 3u || ch == 'u');
		}
		lo[a][NCH] = hi[a][NCH] = 0;
		inv[a][NCH] = invm[a][NCH] = ~0ull;
		if (lane == 0) {
			const lds_ptr<uint64_t> k = a ? kpl1 : kpl0;
#pragma unroll
			for (int c = 0; c <= NCH; ++c) {
#if URX_SLOT_STREAM
				// the two letter planes as ONE interleaved stream (dev_common.h: kmer_slots): word 2c = bases 64c .. 64c + 31, bit 2i = low letter bit of base i,
				// bit 2i + 1 = high bit -- what slot_from_planes cuts a k-mer's 2W bits out of (the spreads are the ones kmer_slots makes of the same ballots)
				k[2 * c] = spread32(lo[a][c]) | (spread32(hi[a][c]) << 1);
				k[2 * c + 1] = spread32(lo[a][c] >> 32) | (spread32(hi[a][c] >> 32) << 1);
#else
				k[c] = lo[a][c]; k[(NCH + 1) + c] = hi[a][c];
#endif
				k[2 * (NCH + 1) + c] = inv[a][c]; k[3 * (NCH + 1) + c] = invm[a][c];
			}
		}
	}
	uint64_t sp[2][NCH], sm[2][NCH];
	uint32_t rp[2][NCH][2], rm[2][NCH][2];
	bool vp[2][NCH], vm[2][NCH];
#pragma unroll
	for (int a = 0; a < 2; ++a) {
		const uint32_t nwords = QL[a] - (W - 1);
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			sp[a][c] = sm[a][c] = 0; vp[a][c] = vm[a][c] = false;
			if (64u * c < nwords)  // wave-uniform
				kmer_slots(X, lo[a][c], hi[a][c], inv[a][c], invm[a][c], lo[a][c + 1], hi[a][c + 1], inv[a][c + 1], invm[a][c + 1], lane,
				           64u * c + lane, nwords, sp[a][c], sm[a][c], vp[a][c], vm[a][c]);
		}
	}
#pragma unroll
	for (int a = 0; a < 2; ++a) {
		const uint32_t nwords = QL[a] - (W - 1);
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			rp[a][c][0] = rp[a][c][1] = rm[a][c][0] = rm[a][c][1] = 0;
			if (64u * c < nwords && X.slot16) {
				// round 5: the 16-byte slots of DevIndex::slot16 -- aligned (never two sectors), position and tally in their first two
				// dwords without the 5-byte unpack.  (A head behind a long link shows its row's resolved first position there, not the
				// link's steps: the seed enumeration reads positions of BOTH1 slots only.)
				const uint2 qp = *reinterpret_cast<const uint2 *>(X.slot16 + (vp[a][c] ? sp[a][c] : 0ull));
				const uint2 qm = *reinterpret_cast<const uint2 *>(X.slot16 + (vm[a][c] ? sm[a][c] : 0ull));
				rp[a][c][0] = qp.x; rp[a][c][1] = qp.y;
				rm[a][c][0] = qm.x; rm[a][c][1] = qm.y;
			} else if (64u * c < nwords) {  // lanes without a valid word fetch slot 0 (one cached sector) and drop it
				const uint64_t ap = vp[a][c] ? 5ull * sp[a][c] : 0ull, am = vm[a][c] ? 5ull * sm[a][c] : 0ull;
				const uint32_t *qp = reinterpret_cast<const uint32_t *>(X.blob + (ap & ~3ull));
				const uint32_t *qm = reinterpret_cast<const uint32_t *>(X.blob + (am & ~3ull));
				rp[a][c][0] = qp[0]; rp[a][c][1] = qp[1];
				rm[a][c][0] = qm[0]; rm[a][c][1] = qm[1];
			}
		}
	}
#pragma unroll
	for (int a = 0; a < 2; ++a) {
		const uint32_t nwords = QL[a] - (W - 1);
#pragma unroll
		for (int c = 0; c < NCH; ++c) {
			const uint32_t p = 64u * c + lane;
			if (p < nwords) {
				// (tally in the low byte, position above it: slot16's {position, tally | ...} is brought into the 5-byte slot's shape)
				const uint64_t xp = X.slot16 ? (((uint64_t)rp[a][c][0] << 8) | (rp[a][c][1] & 0xFFu))
				                             : (((uint64_t)rp[a][c][1] << 32) | rp[a][c][0]) >> (8u * (uint32_t)((5ull * sp[a][c]) & 3ull));
				const uint64_t xm = X.slot16 ? (((uint64_t)rm[a][c][0] << 8) | (rm[a][c][1] & 0xFFu))
				                             : (((uint64_t)rm[a][c][1] << 32) | rm[a][c][0]) >> (8u * (uint32_t)((5ull * sm[a][c]) & 3ull));
				const uint32_t pm = QL[a] - W - p;  // minus-strand position of the k-mer over the same bases
				const uint8_t tp = vp[a][c] ? (uint8_t)(xp & 0xFF) : (uint8_t)TALLY_FREE, tm = vm[a][c] ? (uint8_t)(xm & 0xFF) : (uint8_t)TALLY_FREE;
				const uint32_t pp = vp[a][c] ? (uint32_t)(xp >> 8) : 0xFFFFFFFFu, pmn = vm[a][c] ? (uint32_t)(xm >> 8) : 0xFFFFFFFFu;
				s_tal[a][0][p] = tp; s_pos[a][0][p] = pp;
				s_tal[a][1][pm] = tm; s_pos[a][1][pm] = pmn;
			} else if (p < QL[a]) {
				s_tal[a][0][p] = 0; s_pos[a][0][p] = 0xFFFFFFFFu;
				s_tal[a][1][p] = 0; s_pos[a][1][p] = 0xFFFFFFFFu;
			}
		}
	}
}

// The slot of the k-mer at query position q of strand s (0 plus, 1 minus) of a read whose letter planes probe_pair left in
// LDS (kpl: [plane][NCHP1] words, planes lo / hi / inv / invm), each lane its own (s, q).  ~0 where there is no k-mer
// (a letter outside ACGTU in it, or q beyond the last k-mer start) -- State1::SetSlotsVec's UINT64_MAX (state1.cpp:396-438).
template <int NCHP1>
__device__ __forceinline__ uint64_t slot_from_planes(const DevIndex &X, lds_ptr<const uint64_t> kpl, uint32_t nwords, int s, uint32_t q) {
	const uint32_t W = X.W;
	if (q >= nwords) return ~0ull;
	const uint32_t p = s ? nwords - 1u - q : q;  // the plus-strand position of the k-mer's first base
	const uint32_t c = p >> 6, sh = p & 63u;
	const uint64_t wmask = (W >= 32) ? 0xFFFFFFFFull : ((1ull << W) - 1ull);
	auto cut = [&](int plane) -> uint64_t {
		const uint64_t w0 = kpl[plane * NCHP1 + c], w1 = kpl[plane * NCHP1 + c + 1];  // c + 1 <= NCH: the last word is the pad
		uint64_t f = w0 >> sh;
		if (sh) f |= w1 << (64u - sh);
		return f & wmask;
	};
	const uint64_t fbad = cut(s ? 3 : 2);
	if (fbad) return ~0ull;
	uint64_t w;
#if URX_SLOT_STREAM
	// 2W bits of the interleaved stream from bit 2p (words 2c, 2c + 1 of a chunk: index p >> 5)
	const uint32_t wi = p >> 5, sh2 = (2u * p) & 63u;
	uint64_t seg = kpl[wi] >> sh2;
	if (sh2) seg |= kpl[wi + 1] << (64u - sh2);  // wi + 1 <= 2 NCH: the pad chunk's words
	const uint64_t m2 = (W >= 32) ? ~0ull : ((1ull << (2u * W)) - 1ull);
	if (s == 0) {  // first base most significant: the letters' bits swapped, then the piece bit-reversed
		const uint64_t t = ((seg & 0x5555555555555555ull) << 1) | ((seg >> 1) & 0x5555555555555555ull);
		w = __brevll(t & m2) >> (64u - 2u * W);
	} else
		w = ~seg & m2;
#else
	const uint64_t flo = cut(0), fhi = cut(1);
	if (s == 0) w = spread32(__brevll(flo) >> (64 - W)) | (spread32(__brevll(fhi) >> (64 - W)) << 1);
	else w = spread32(~flo & wmask) | (spread32(~fhi & wmask) << 1);
#endif
	return mod_slots(murmur64(w & X.shiftMask), X.slotCount, X.slotMagic);
}

}  // namespace urx
