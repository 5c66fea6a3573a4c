// k_tb.hip -- "time to backup set" implicit filter: ASIFimplicitTB::filter, src/asif_implicit_tb.cpp:261-363.
//
// Stage 1  tb_rows_kernel, one instance per lane:
//   * backupSet(x) >= 0  -> trivial rows A = 0, b = -inf (updateConstraintsTrivial, :716-733), code 2;
//   * otherwise forward-Euler integration of the backup closed loop + sensitivity (:464-488) until the
//     first sample inside the backup set ("first hit wins", :505-528 -- nothing past that sample is
//     ever read by the reference, so the lane stops integrating there); no hit within the horizon
//     -> code -3 (:529-536);
//   * rows (:539-674): the npBTSS most critical samples of [0, idxHit] (padded with inert rows h=1,
//     Dh=0 when the trajectory is shorter), the time-to-safety row and the orthogonality row with its
//     analytic gradient through the hitting map.
//   Lanes of a wave stop at different samples; the loop runs to the wave's slowest lane.
// Stage 2  qp_policy_kernel<2,18,G> with the TB epilogue: rc 2 / 1 / -1 / raw solver status / -3 and the
//   saturated backup controller on every failure (:290-361).
#include <cstdlib>
#include <type_traits>
#include "backup_traj.hpp"
#include "qp_kernel.hpp"

namespace asif {


// ---- pass 1 in two roles (models that declare kTbSplitRoles: the segway) ------------------------------------------
// One wave per 64 instances runs at the issue rate of ONE instruction stream, and at the BASELINE batch (32 768
// instances per GPU = 512 waves) half of the chip's 1 024 SIMDs hold no wave at all.  The state x of the backup
// trajectory does not depend on its sensitivity Q; measured on C4, integrating x alone takes 0.19 ms of the fused pass's
// 0.335.  So the pass is dealt to TWO waves of one workgroup, on two SIMDs: wave 0 integrates x (controller, saturation,
// f, g, margins, hit test, the running selection of critical blocks) and leaves a record of every step in an LDS ring;
// wave 1, one block of MB steps behind, evaluates the gradients at the recorded states and integrates Q.  A workgroup
// barrier per block hands a ring buffer over (two buffers: the x wave fills one while the Q wave reads the other).
// Block checkpoints: the x wave decides which block enters the selection and stores the x part; the slot travels with
// the record and the Q wave stores its part.  After the pass the Q wave publishes its final Q and retires; the x wave
// goes on alone with pass 2 and the rows exactly as the fused kernel (same code below).  A range / NaN alarm from either
// role sends the x wave through the fused, checking pass instead.  Taken only when the grid is small enough that the
// second wave has a SIMD to itself (two workgroups of 77 KB per CU); larger batches run the fused kernel.
// models that declare kTbFuseQp have the rows kernel solve the instance's QP itself (the kernel's last block); worth it
// where stage 2 is a visible share of the step (segway 7 %, double integrator 5 %) -- on the pendulum's 11 551-step pass
// it is 0.7 %, and the solver's registers in the kernel cost the pass's loop 3 %
template <class M, class = void>
struct tb_fuse_qp : std::false_type {};
template <class M>
struct tb_fuse_qp<M, std::enable_if_t<M::kTbFuseQp>> : std::true_type {};
template <class M, class = void>
struct tb_split_roles : std::false_type {};
template <class M>
struct tb_split_roles<M, std::enable_if_t<M::kTbSplitRoles>> : std::true_type {};

template <class M>
struct TbSplitLds {
	static constexpr int NX = M::NX, NZ = NX + NX * NX, K = M::NPBTSS, MB = M::kTrajBlock, NF = BackupLoop<M>::kRecordDoubles;
	static constexpr int kSlots = K * NZ * 64;          // checkpoints (pass 1), payload (pass 2): doubles
	static constexpr int kRing = 2 * MB * NF * 64;      // records of two blocks of steps: doubles
	static constexpr int kCtl = 2 * MB * 64 + 64 + 4;   // per-step control words, last-block slots, stop / alarm flags: ints
	static constexpr size_t bytes() { return sizeof(double) * (kSlots + kRing) + sizeof(int) * kCtl; }
};

// The Q wave of the two-role pass (see TbSplitLds above): one block of steps behind the x wave, it evaluates the
// gradients at the recorded states and integrates the sensitivity.  Barriers: one per block (A_b), one after the x
// wave has published every lane's last-block slot (B1), one after this wave has published its final Q (B2) -- the x
// wave executes exactly the same sequence.
template <class M>
__device__ __forceinline__ void tb_q_role(const DevOptions &o, const double *ring, const int *ctl, const int *lastSlot,
                                          int *flags, double *ckl, double *qout, int lane, int npBT)
{
	using BL = BackupLoop<M>;
	constexpr int NX = M::NX, NZ = NX + NX * NX, MB = M::kTrajBlock, NF = BL::kRecordDoubles;
	double q[NX * NX], qs[NX * NX]; // Q now, and at the first sample of the current block
#pragma unroll
	for (int e = 0; e < NX * NX; e++) q[e] = qs[e] = (e % (NX + 1) == 0) ? 1.0 : 0.0;
	const int nblk = (npBT + MB - 1) / MB;
	constexpr unsigned kPinQ = M::kPinQ;
	const typename M::Consts kq = M::template constants<kPinQ>(); // made once: the pinned groups live in VGPRs from here on
	auto keep = [&](int slot) { // Q part of the checkpoint of the block that just closed
#pragma unroll
		for (int e = 0; e < NX * NX; e++) ckl[(slot * NZ + NX + e) * 64 + lane] = qs[e];
	};
	auto record = [&](int buf, int k) { // the x wave's record of step k of the block in ring buffer `buf`
		typename BL::StepRecord r;
		const double *rec = ring + (size_t)((buf * MB + k) * NF) * 64 + lane;
		r.xg[0] = rec[0 * 64];
		r.xg[1] = rec[1 * 64];
		r.uSat = rec[2 * 64];
		r.DuSat = rec[3 * 64];
		r.s = rec[4 * 64];
		r.c = rec[5 * 64];
		r.iden = rec[6 * 64];
		r.rg = rec[7 * 64];
		return r;
	};
	// one step of the sensitivity; the step into the first sample of a block closes the block before: its checkpoint's
	// Q part goes to the slot the x wave chose, and the new block's start is noted
	auto step = [&](const typename BL::StepRecord &r, int c, bool blockStart) {
		BL::template stepQ<(kPinQ & M::kPinTanh) != 0>(o, r, q, kq);
		if (blockStart) {
			const int slot = (c >> 8) - 1;
			if (slot >= 0) keep(slot);
#pragma unroll
			for (int e = 0; e < NX * NX; e++) qs[e] = q[e];
		}
	};
	// A lane that did not step (it reached the backup set, or started inside it: a third of the segway's seeded batch)
	// holds a stale record and sits the step out under the exec mask: one s_and_saveexec + skip branch per step where
	// "compute and drop" paid two selects per entry of Q.  Whether the sample exists at all (the first block starts at
	// sample 1, the last may be partial) is wave-uniform and goes into the same mask instead of a branch of its own -- a
	// lone wave pays ~50 cycles for every branch instruction, taken or not.  Nothing hides an LDS round trip from a lone
	// wave either: the block's control words are read together, and the record of step k + 1 is on its way while step k
	// computes (the loads are unconditional -- a stale record is read and not used).
#pragma unroll 1
	for (int b = 0; b < nblk; b++) {
		__syncthreads(); // A_b: buffer b & 1 holds block b (or the stop flag)
		const int buf = b & 1;
		if (flags[buf]) break; // wave-uniform
		int c[MB];
#pragma unroll
		for (int k = 0; k < MB; k++) c[k] = ctl[(buf * MB + k) * 64 + lane]; // (stale where the sample does not exist)
		typename BL::StepRecord rn = record(buf, 0);
#pragma unroll
		for (int k = 0; k < MB; k++) {
			const int s = b * MB + k;
			const typename BL::StepRecord r = rn;
			if (k + 1 < MB) rn = record(buf, k + 1);
			if ((s != 0) & (s < npBT) & ((c[k] & 1) != 0)) step(r, c[k], k == 0 && b > 0);
		}
	}
	__syncthreads(); // B1: every lane's last-block slot is published
	{
		const int slot = lastSlot[lane];
		if (slot >= 0) keep(slot);
	}
	bool bad = false;
#pragma unroll
	for (int e = 0; e < NX * NX; e++) {
		qout[e * 64 + lane] = q[e];
		bad = bad || (q[e] != q[e]);
	}
	if (__any(bad) && lane == 0) flags[2] = 1;
	__syncthreads(); // B2: final Q and the alarm flag are published
}

// CKPT: where the block-start states of pass 1's running selection live (K slots of NZ doubles per lane).
//   kCkptLds2  in LDS, in a region of their own next to pass 2's payload (2 x K NZ 64 doubles = 80 KB for the segway:
//              two waves per CU) -- nothing of the search ever touches HBM; taken when the batch needs no more than
//              two waves per CU anyway (C4: 32 768 instances per GPU = 512 waves on 256 CUs);
//   kCkptSpill in LDS during pass 1, in the region pass 2 will use for its payload; the K survivors are written to
//              HBM once, between the passes (640 B per instance), and pass 2 reads them back.
// Round 2 wrote a checkpoint to HBM whenever a block ENTERED the selection: on the segway, whose margins shrink along
// the trajectory, nearly every 4-sample block does -- 171 MB of writes per 32 768 instances (PMC), 5.2 KB per instance.
constexpr int kCkptLds2 = 1, kCkptSpill = 2;
// a.code of an instance whose QP the rows kernel left to stage 2 (fused mode): code + this (codes are -3, 1, 2)
constexpr int kTbPendingMark = 8;
// waves per workgroup of the fused pass that the kernel is compiled for: the two-state models' kernels (200 VGPRs) take
// up to four; the segway's sits at the 256-VGPR line, and a larger launch bound makes the compiler cross it further
// (266 with AGPRs against 262)
template <class M>
constexpr int tb_max_wg_waves() { return M::NX <= 2 ? 4 : 1; }

template <class M, int CKPT, bool SPLIT = false, bool DOPRI = false>
__global__ __launch_bounds__(SPLIT ? 128 : 64 * tb_max_wg_waves<M>()) void tb_rows_kernel(DevOptions o_arg, FilterArgs a)
{
	static_assert(!SPLIT || CKPT == kCkptSpill, "the two-role pass keeps its step records where the own-region checkpoints would be");
	// the soft saturation selects between these two and the input: as kernel arguments (SGPRs) they are copied into
	// VGPRs at every Euler step; an opaque copy made once, here, stays in two VGPR pairs for the whole kernel
	DevOptions o = o_arg;
	asm("" : "+v"(o.lb[0]), "+v"(o.ub[0]));
	// two-role pass: the scalars the saturation and the step read at every sample, likewise -- the scalar file does not
	// hold them next to the masks of the x role's per-sample tests and they came back from spill lanes (v_readlane), 18
	// per step
	if constexpr (SPLIT)
		asm("" : "+v"(o.satMiddle), "+v"(o.twoOverRange), "+v"(o.bevelStop), "+v"(o.bevelStart), "+v"(o.satSharpness),
		    "+v"(o.satRange), "+v"(o.trajDt));
	constexpr int NX = M::NX, NP = M::NPSS, K = M::NPBTSS, NZ = NX + NX * NX;
	constexpr int NC = K * NP + 2, NV = 2;
	extern __shared__ double tb_lds_all[];
	double *const tb_lds = tb_lds_all + (SPLIT ? 0 : (size_t)(threadIdx.x >> 6) * (CKPT == kCkptLds2 ? 2 : 1) * K * NZ * 64);
	double *const pay = tb_lds;                                            // pass 2: states of the K most critical samples
	double *const ckl = CKPT == kCkptLds2 ? tb_lds + K * NZ * 64 : tb_lds; // pass 1: states at the start of the K selected blocks
	// fused pass: a workgroup is one to four waves that share nothing (launchers.hpp: waves_per_workgroup), each with LDS
	// regions of its own; two-role pass: the two waves of one 64-instance group
	const int lane = (int)(threadIdx.x & 63);
	const int wv = SPLIT ? 0 : (int)(threadIdx.x >> 6), nwv = SPLIT ? 1 : (int)(blockDim.x >> 6);
	int64_t i = ((int64_t)blockIdx.x * nwv + wv) * 64 + lane;
	if (!SPLIT && i - lane >= a.B) return; // a wave past the end of the batch (wave-uniform; the fused pass has no barrier)
	const bool live = i < a.B;
	if (!live) i = a.B - 1;
	const int64_t ld = a.ld;
	// two-role pass: step records, per-step control words, last-block slots, flags -- behind the checkpoint region
	using SL = TbSplitLds<M>;
	double *const ring = tb_lds + SL::kSlots;
	int *const ctl = (int *)(ring + SL::kRing);
	int *const lastSlot = ctl + 2 * M::kTrajBlock * 64;
	int *const flags = lastSlot + 64; // [0], [1]: "every lane was done before the block in this buffer"; [2]: alarm from the Q wave
	if constexpr (SPLIT) {
		if (threadIdx.x == 0) flags[2] = 0;
		if (threadIdx.x >= 64) { // wave 1: the sensitivity
			// (options outside the fast step's preconditions: the x wave runs the fused, checking pass on its own and
			// never reaches a barrier -- the launcher does not pick this kernel then, and this keeps the two in step anyway)
			if (o.satFastOk) tb_q_role<M>(o, ring, ctl, lastSlot, flags, ckl, ring, lane, o.npBT);
			return;
		}
	}

	double x0[NX], f0[NX], g0[NX];
#pragma unroll
	for (int k = 0; k < NX; k++) x0[k] = a.x[k * ld + i];
	const bool inside = M::backupSetValue(o, x0) >= 0; // :278-290
	M::dynamics(o, x0, f0, g0);

	double z[NZ];
#pragma unroll
	for (int k = 0; k < NZ; k++) z[k] = 0.0;
#pragma unroll
	for (int k = 0; k < NX; k++) {
		z[k] = x0[k];
		z[NX + k * (NX + 1)] = 1.0;
	}
	// ---- pass 1: integrate to the first sample inside the backup set, nothing kept per sample.  Per block of MB
	// samples the state at its first sample goes to HBM and the block's smallest margin feeds a running selection of
	// the K blocks with the smallest minima (ties -> earlier block): the K most critical samples of [0, idxHit]
	// all lie in those blocks (see k_implicit.hip).  The per-sample selection this replaces (branch, K-entry
	// network, NZ LDS writes) ran on most steps: 10 % of the segway kernel, 24 % of the pendulum's.
	// The block-start state rides in registers over the block and is stored only when the block enters the selection,
	// into the displaced entry's slot: a.ckpt is [K][NZ][ld] (the segway's 20-double state included: it fits since
	// the model's arithmetic was shortened; one checkpoint per block was 414 MB of writes per launch at 32 768).
	constexpr int MB = M::kTrajBlock;
	TopK<K> topB;
	double *ck = a.ckpt + i;
	double zs[NZ]; // state at the first sample of the current block
	RunningMargin<M> brun; // smallest margin of the current block
	double hall; // smallest safety margin of the whole pass (kTrigUnchecked: it bounds the trig arguments)
	auto commit = [&](int blk) { // close block blk
		const double bmin = brun.value();
		hall = fmin(hall, bmin);
		if (__any(bmin < topB.key[K - 1])) {
			const int slot = topB.insert(bmin, blk);
			if (slot >= 0) {
#pragma unroll
				for (int k = 0; k < NZ; k++) ckl[(slot * NZ + k) * 64 + lane] = zs[k];
			}
		}
	};
	bool done, hit;
	int idxHit, sLast;
	double t, tHit;
	// The whole pass runs on the branch-free trig fast path; its argument range is policed once, after the pass: by
	// the smallest safety margin seen (kTrigUnchecked: these models' margins bound the trig arguments, models.hpp) or
	// by NaN poisoning (kTrigPoison: an argument outside the range turns the lane's state into NaN -- NaN never enters
	// the backup set).  A lane out of range has the pass repeated with the checking version.
	// the fast step's trig mode (models.hpp): carried along the steps, bounded by the margin, or poisoning.  Carried:
	// ONE form of the step (always the rotation); sin / cos are re-synchronised with a fresh evaluation where a block
	// starts -- here at the block boundary, in pass 2 after loading a block's checkpoint: the same arithmetic from the
	// same state, so both passes see the same bits.  (Choosing between two forms of the step inside this loop, with its
	// per-lane exits at every sample, doubled its body and cost the pendulum's 11 551-step pass 10 %.)
	constexpr int kFastTrig = trig_carry<M>::value ? kTrigCarried : (trig_by_margin<M>::value ? kTrigUnchecked : kTrigPoison);
	TrigCarry carry = {0.0, 0.0, 1.0};
	auto resync = [&]() {
		if constexpr (trig_carry<M>::value) {
			sincos_fast<kTrigUnchecked>(z[M::kTrigAngle], carry.s, carry.c);
			carry.x = z[M::kTrigAngle];
		}
	};
	auto pass1 = [&](auto fast) {
		constexpr int P = !decltype(fast)::value ? kTrigChecked : kFastTrig;
#pragma unroll
		for (int k = 0; k < NZ; k++) z[k] = 0.0;
#pragma unroll
		for (int k = 0; k < NX; k++) {
			z[k] = x0[k];
			z[NX + k * (NX + 1)] = 1.0;
		}
		topB.init();
#pragma unroll
		for (int k = 0; k < NZ; k++) zs[k] = z[k]; // block 0 starts at sample 0
		if (P == kTrigCarried) resync();
		brun.reset();
		brun.add(o, x0);
		hall = __builtin_huge_val();
		done = inside || !live;
		hit = false;
		idxHit = 0;
		sLast = 0;
		t = 0.0;
		tHit = 0.0;
		typename BackupLoop<M>::Hold none = {0.0, 0.0};
		// one sample: the step into it, the block bookkeeping where a block starts, margin and hit test; per-lane
		// (lanes that reached the backup set sit out)
		bool bevelSeen = false; // bevel-free steps: a lane met a bevel after all
		auto sample = [&](int s, bool opens, auto nobevel) {
			constexpr bool NB = decltype(nobevel)::value;
			if (!done) {
				if constexpr (tb_split_roles<M>::value && P == kTrigCarried) BackupLoop<M>::eulerStepRoles(o, z, carry);
				else BackupLoop<M>::template eulerStepT<false, P, NB>(o, z, none, 0.0, &carry, false, &bevelSeen);
				t = t + o.trajDt; // backTraj_[i].first accumulates, :475
				sLast = s;
				if (opens) { // wave-uniform: close the previous block, open the next
					commit(s / MB - 1);
					brun.reset();
#pragma unroll
					for (int k = 0; k < NZ; k++) zs[k] = z[k];
					if (P == kTrigCarried) resync();
				}
				double xs[NX];
#pragma unroll
				for (int k = 0; k < NX; k++) xs[k] = z[k];
				brun.add(o, xs);
				if (M::backupSetInside(o, xs)) {
					hit = true;
					done = true;
					idxHit = s;
					tHit = t;
				}
			}
		};
		if constexpr (M::kTbUnrollSteps) {
			// A small step (the pendulum's ~100 instructions): block by block -- whether every lane is done is asked once
			// per block, and full blocks have a compile-time trip count, unrolled by four (the loop's own control is scalar
			// work that takes issue turns from the one wave on the SIMD): 3.13 -> 2.64 ms on the 11 551-step pendulum.
#pragma unroll 1
			for (int s0 = 0; s0 < o.npBT; s0 += MB) {
				if (__all(done)) break;
				if (s0 > 0 && s0 + MB <= o.npBT) {
					// As in k_implicit.hip: a block none of whose lanes starts near a bevel of the soft saturation runs the
					// step without the bevel code and its branch (bevel_rate, models.hpp).  The sample that opens the block
					// takes the full step; from there the block's start state is zs, so a block that met a bevel after all
					// (bevelSeen) goes back to it and runs the full step.
					bool far = false;
					if constexpr (bevel_rate<M>::value > 0.0 && P != kTrigChecked) {
						sample(s0, true, std::false_type());
						if (o.bevelFree) {
							double xs[NX], u0[1], Du0[NX];
#pragma unroll
							for (int c = 0; c < NX; c++) xs[c] = z[c];
							M::backupController(o, xs, u0, Du0);
							const double au = fabs((u0[0] - o.satMiddle) * o.twoOverRange);
							const double d = o.bevelFree == 2 ? 0.0 : bevel_rate<M>::value * (double)MB * o.trajDt;
							far = !__any(!done && ((au > o.bevelStart - d && au < o.bevelStop + d) || (o.bevelFree != 2 && !M::arrivalFar(o, xs, MB))));
						}
						if (far) {
							const double t_b = t, tHit_b = tHit;
							const int sLast_b = sLast, idxHit_b = idxHit;
							const bool done_b = done, hit_b = hit;
							bevelSeen = false;
							// ... and without the per-sample questions "is this lane still integrating" and "has it reached the
							// backup set": the first cannot change inside a block in which no lane arrives, so it is asked once
							// for the block, and an arrival (each lane has one in its life) is latched like a bevel and the block
							// repeated with the full sample.  Two divergent regions and their branches less per step.
							bool hitSeen = false;
							if (!done) {
								auto quiet = [&]() {
									BackupLoop<M>::template eulerStepT<false, P, true>(o, z, none, 0.0, &carry, false, &bevelSeen);
									t = t + o.trajDt;
									double xs[NX];
#pragma unroll
									for (int c = 0; c < NX; c++) xs[c] = z[c];
									brun.add(o, xs);
									hitSeen = hitSeen || M::backupSetInside(o, xs);
								};
								if constexpr (M::kTbUnrollWholeBlock) {
#pragma unroll
									for (int k = 1; k < MB; k++) quiet();
								} else {
#pragma unroll 4
									for (int k = 1; k < MB; k++) quiet();
								}
								sLast = s0 + MB - 1;
							}
							if (__any(bevelSeen || hitSeen)) { // mispredicted: the block again from its first sample
								far = false;
								if (!done_b) { // (a lane that had reached the backup set before keeps the state it stopped in)
#pragma unroll
									for (int c = 0; c < NZ; c++) z[c] = zs[c];
									if (P == kTrigCarried) resync();
									double xs[NX];
#pragma unroll
									for (int c = 0; c < NX; c++) xs[c] = z[c];
									brun.reset();
									brun.add(o, xs);
									t = t_b;
									tHit = tHit_b;
									sLast = sLast_b;
									idxHit = idxHit_b;
									done = done_b;
									hit = hit_b;
								}
							}
						}
						if (!far) {
							if constexpr (M::kTbUnrollWholeBlock) {
#pragma unroll
								for (int k = 1; k < MB; k++) sample(s0 + k, false, std::false_type());
							} else {
#pragma unroll 4
								for (int k = 1; k < MB; k++) sample(s0 + k, false, std::false_type());
							}
						}
					} else if constexpr (M::kTbUnrollWholeBlock) {
#pragma unroll
						for (int k = 0; k < MB; k++) sample(s0 + k, k == 0, std::false_type());
					} else {
#pragma unroll 4
						for (int k = 0; k < MB; k++) sample(s0 + k, k == 0, std::false_type());
					}
				} else {
#pragma unroll 1
					for (int k = (s0 == 0 ? 1 : 0); k < MB && s0 + k < o.npBT; k++) sample(s0 + k, k == 0 && s0 > 0, std::false_type());
				}
			}
		} else {
			// The segway's ~500-instruction step: one copy of the step in one loop (the block form, even without
			// unrolling, measured 6 % slower there)
#pragma unroll 1
			for (int s = 1; s < o.npBT; s++) {
				if (__all(done)) break;
				sample(s, s % MB == 0, std::false_type());
			}
		}
		commit(sLast / MB); // every lane's last (possibly partial) block
	};
	// the x wave of the two-role pass: pass1(fast) without the sensitivity, a record per step for the Q wave
	bool alarm = false;
	auto pass1x = [&]() {
	  if constexpr (SPLIT) {
		using BL = BackupLoop<M>;
		constexpr int NF = BL::kRecordDoubles;
#pragma unroll
		for (int k = 0; k < NZ; k++) z[k] = 0.0;
#pragma unroll
		for (int k = 0; k < NX; k++) {
			z[k] = x0[k];
			z[NX + k * (NX + 1)] = 1.0;
		}
		topB.init();
#pragma unroll
		for (int k = 0; k < NZ; k++) zs[k] = z[k];
		resync();
		brun.reset();
		brun.add(o, x0);
		hall = __builtin_huge_val();
		done = inside || !live;
		hit = false;
		idxHit = 0;
		sLast = 0;
		t = 0.0;
		tHit = 0.0;
		auto commitX = [&](int blk) { // as commit(), the x part of the checkpoint; the slot goes to the Q wave
			int slot = -1;
			const double bmin = brun.value();
			hall = fmin(hall, bmin);
			if (__any(bmin < topB.key[K - 1])) {
				slot = topB.insert(bmin, blk);
				if (slot >= 0) {
#pragma unroll
					for (int k = 0; k < NX; k++) ckl[(slot * NZ + k) * 64 + lane] = zs[k];
				}
			}
			return slot;
		};
		const int nblk = (o.npBT + MB - 1) / MB;
		constexpr unsigned kPinX = M::kPinX;
		const typename M::Consts kx = M::template constants<kPinX>(); // made once: the pinned groups live in VGPRs from here on
#pragma unroll 1
		for (int b = 0; b < nblk; b++) {
			const int buf = b & 1;
			const bool stop = __all(done);
			if (lane == 0) flags[buf] = stop ? 1 : 0;
			if (!stop) {
#pragma unroll
				for (int k = 0; k < MB; k++) { // unrolled: the record's LDS addresses are a base per buffer plus immediates
					const int s = b * MB + k;
					// (whether the sample exists -- wave-uniform -- in the lanes' mask rather than a branch of its own)
					int c = 0;
					if ((s != 0) & (s < o.npBT) & !done) {
						double xs[NX];
#pragma unroll
						for (int e = 0; e < NX; e++) xs[e] = z[e];
						const typename BL::StepRecord r = BL::template stepX<(kPinX & M::kPinCarry) != 0>(o, xs, carry, kx);
#pragma unroll
						for (int e = 0; e < NX; e++) z[e] = xs[e];
						t = t + o.trajDt;
						sLast = s;
						int slot = -1;
						if (k == 0 && b > 0) { // wave-uniform: close the previous block, open this one
							slot = commitX(b - 1);
							brun.reset();
#pragma unroll
							for (int e = 0; e < NX; e++) zs[e] = z[e];
							resync();
						}
						brun.add(o, xs);
						if (M::backupSetInside(o, xs)) {
							hit = true;
							done = true;
							idxHit = s;
							tHit = t;
						}
						double *rec = ring + (size_t)((buf * MB + k) * NF) * 64 + lane;
						rec[0 * 64] = r.xg[0];
						rec[1 * 64] = r.xg[1];
						rec[2 * 64] = r.uSat;
						rec[3 * 64] = r.DuSat;
						rec[4 * 64] = r.s;
						rec[5 * 64] = r.c;
						rec[6 * 64] = r.iden;
						rec[7 * 64] = r.rg;
						c = 1 | ((slot + 1) << 8);
					}
					ctl[(buf * MB + k) * 64 + lane] = c;
				}
			}
			__syncthreads(); // A_b
			if (stop) break;
		}
		lastSlot[lane] = commitX(sLast / MB); // every lane's last (possibly partial) block
		__syncthreads(); // B1
		__syncthreads(); // B2: the Q wave's final Q sits where the records were
#pragma unroll
		for (int e = 0; e < NX * NX; e++) z[NX + e] = ring[e * 64 + lane];
		alarm = flags[2] != 0;
	  }
	};
	double zHit[NZ]; // the rows below are written for the state at idxHit
	TopK<K> top;     // the K most critical samples of [0, idxHit], their states in LDS (payp)
	top.init();
	// Two-role pass: nothing is spilled.  Its step records are history by now, so K - 1 payload entries go where the
	// ring was; the K-th goes into the slot of the checkpoint that pass 2 consumes FIRST (it is in registers before any
	// payload is stored), and the other checkpoints stay in LDS until their turn.  (42 MB of HBM traffic per 32 768
	// instances less than writing the survivors out and reading them back: 74 -> 32 MB.)
	int slFirst = 0;
	static_assert(!SPLIT || (K - 1) * NZ * 64 <= SL::kRing, "K - 1 payload entries fit the record ring");
	auto payp = [&](int slot) -> double * { // entry `slot` of the payload, this lane's column
		if constexpr (SPLIT) return (slot < K - 1 ? ring + slot * NZ * 64 : ckl + slFirst * NZ * 64) + lane;
		else return pay + slot * NZ * 64 + lane;
	};
	if constexpr (DOPRI) {
		// The reference's USE_ODEINT build (src/asif_implicit_tb.cpp:431-463): the samples are the dense output of an
		// adaptive dopri5 at t = i backTrajDt instead of Euler steps.  The adaptive step straddles samples, so there is no
		// per-block restart point: ONE pass with the exact per-sample selection, states parked in LDS; a lane stops at its
		// first sample inside the backup set (nothing past it is ever read, :505-528).  Unpinned like the implicit class's
		// (Boost absent; the controller is restated in oracle/or_assembly.c).
		Dopri5<M> rk;
		rk.init(o, z, o.trajDt);
		done = inside || !live;
		hit = false;
		idxHit = 0;
		sLast = 0;
		t = 0.0;
		tHit = 0.0;
#pragma unroll
		for (int k = 0; k < NZ; k++) zHit[k] = z[k];
		{
			const int slot = top.insert(M::safetyMin(o, x0), 0); // sample 0 (never a hit: `inside` is decided above)
			if (slot >= 0) {
				double *pp = payp(slot);
#pragma unroll
				for (int k = 0; k < NZ; k++) pp[k * 64] = z[k];
			}
		}
		int guard = 200000; // as in k_implicit.hip: a stuck controller must not hang a wave
#pragma unroll 1
		for (int s = 1; s < o.npBT; s++) {
			if (__all(done)) break;
			const double ts = o.trajDt * (double)s; // backTraj_[i].first, :451
			while (guard > 0 && __any(!done && rk.behind(ts))) {
				rk.tryStep(o, !done && rk.behind(ts));
				guard--;
			}
			rk.failed = rk.failed | (!done && rk.behind(ts));
			double zs[NZ], xs[NX];
			rk.dense(ts, zs);
#pragma unroll
			for (int k = 0; k < NX; k++) xs[k] = zs[k];
			const double hm = done ? __builtin_huge_val() : M::safetyMin(o, xs);
			if (__any(hm < top.key[K - 1])) {
				const int slot = top.insert(hm, s);
				if (slot >= 0) {
					double *pp = payp(slot);
#pragma unroll
					for (int k = 0; k < NZ; k++) pp[k * 64] = zs[k];
				}
			}
			if (!done) {
				sLast = s;
				t = ts;
				if (M::backupSetInside(o, xs)) {
					hit = true;
					done = true;
					idxHit = s;
					tHit = ts;
#pragma unroll
					for (int k = 0; k < NZ; k++) zHit[k] = zs[k];
				}
			}
		}
	} else {
	// (the fast pass also takes the soft saturation's short forms, valid for ordinary saturation constants --
	// DevOptions::satFastOk, checked on the host; other options run the generic pass)
	bool redo = !o.satFastOk;
	if (!redo) {
		if constexpr (SPLIT) pass1x();
		else pass1(std::true_type());
		bool bad = alarm;
#pragma unroll
		for (int k = 0; k < NZ; k++) bad = bad || (k < NX ? !(fabs(z[k]) < kStateSane) : (z[k] != z[k])); // x: NaN or beyond any sane magnitude; Q: NaN (a stiff model's sensitivity may overflow under forward Euler, as it does upstream)
		if constexpr (trig_carry<M>::value) bad = bad || !M::trigCarryBounded(o, hall);
		else if constexpr (trig_by_margin<M>::value) bad = bad || !M::trigArgsBounded(hall);
		redo = __any(bad); // never on sane trajectories
	}
	if (redo) pass1(std::false_type());
#pragma unroll
	for (int k = 0; k < NZ; k++) zHit[k] = z[k];
	if constexpr (CKPT == kCkptSpill && !SPLIT) {
		// the survivors of the selection leave LDS before pass 2 starts to overwrite the region with its payload
#pragma unroll 1
		for (int p = 0; p < K; p++) {
			int slot = 0, idx = -1;
#pragma unroll
			for (int q = 0; q < K; q++) {
				slot = q == p ? topB.slot[q] : slot;
				idx = q == p ? topB.idx[q] : idx;
			}
			if (idx >= 0) {
				double *c = ck + (int64_t)slot * NZ * ld;
#pragma unroll
				for (int k = 0; k < NZ; k++) c[k * ld] = ckl[(slot * NZ + k) * 64 + lane];
			}
		}
	}

	// ---- pass 2: re-integrate the selected blocks in increasing order, exact per-sample selection with the states
	// parked in LDS; samples beyond the lane's last one (its hit) do not take part
	int cur = -1;
	const bool fast2 = o.satFastOk && !redo; // wave-uniform
#pragma unroll 1
	for (int j = 0; j < K; j++) {
		int nb = 0x7fffffff, sl = 0;
#pragma unroll
		for (int p = 0; p < K; p++) {
			const int v = topB.idx[p];
			const bool better = v > cur && v < nb;
			nb = better ? v : nb;
			sl = better ? topB.slot[p] : sl;
		}
		const bool have = nb != 0x7fffffff;
		if (!__any(have)) break;
		cur = have ? nb : cur;
		const int blk = have ? nb : 0;
		if constexpr (SPLIT) {
			if (j == 0) slFirst = have ? sl : 0;
		}
		if constexpr (CKPT == kCkptSpill && !SPLIT) {
			const double *c = ck + (int64_t)(have ? sl : 0) * NZ * ld;
#pragma unroll
			for (int k = 0; k < NZ; k++) z[k] = c[k * ld];
		} else {
			const int sl2 = have ? sl : 0;
#pragma unroll
			for (int k = 0; k < NZ; k++) z[k] = ckl[(sl2 * NZ + k) * 64 + lane];
		}
		if (fast2) resync(); // the block's first sample: as pass 1 did at this boundary
#pragma unroll 1
		for (int tt = 0; tt < MB; tt++) {
			const int s = blk * MB + tt;
			if (tt > 0) {
				// pass 1 has range-checked these very states (unless it had to be redone): the fast step is valid again
				typename BackupLoop<M>::Hold none2 = {0.0, 0.0};
				if constexpr (tb_split_roles<M>::value && kFastTrig == kTrigCarried) {
					if (fast2) BackupLoop<M>::eulerStepRoles(o, z, carry);
					else BackupLoop<M>::eulerStep(o, z);
				} else {
					if (fast2) BackupLoop<M>::template eulerStepT<false, kFastTrig>(o, z, none2, 0.0, &carry, false);
					else BackupLoop<M>::eulerStep(o, z);
				}
			}
			double xs[NX];
#pragma unroll
			for (int k = 0; k < NX; k++) xs[k] = z[k];
			const double hm = (have && s <= sLast) ? M::safetyMin(o, xs) : __builtin_huge_val();
			if (__any(hm < top.key[K - 1])) {
				const int slot = top.insert(hm, s);
				if (slot >= 0) {
					double *pp = payp(slot);
#pragma unroll
					for (int k = 0; k < NZ; k++) pp[k * 64] = z[k];
				}
			}
		}
	}
	}
#pragma unroll
	for (int k = 0; k < NZ; k++) z[k] = zHit[k];
	if (!live) return;

	const int code = inside ? 2 : (hit ? 1 : -3);
	// The rows of this instance, [Lgh_r, h_r | -Lfh_r], held in registers (every index below is a compile-time one) until
	// the end of the kernel: the filter's own call solves its QP right here (a.fuseQp, below) and nothing is staged;
	// asif_hip_assemble_batch and the other solver modes write them out as before.
	constexpr bool kHold = tb_fuse_qp<M>::value; // (models without the fused solve write each row out as it is made)
	double rA0[kHold ? NC : 1], rA1[kHold ? NC : 1], rB[kHold ? NC : 1];
	auto put = [&](int row, double a0, double a1, double bb) { // row: a compile-time value at every call
		if constexpr (kHold) {
			rA0[row] = a0;
			rA1[row] = a1;
			rB[row] = bb;
		} else {
			a.A[(int64_t)(row + 0 * NC) * ld + i] = a0;
			a.A[(int64_t)(row + 1 * NC) * ld + i] = a1;
			a.b[(int64_t)row * ld + i] = bb;
		}
	};
	if constexpr (!kHold) {
		if (a.code) a.code[i] = code;
	}
	// trivial rows; also what the QP holds for code -3 (the reference leaves A_, b_ untouched there and never solves; an
	// inert QP keeps stage 2 uniform)
	if (kHold || code != 1) {
#pragma unroll
		for (int r = 0; r < NC; r++) put(r, 0.0, 0.0, -o.inf);
	}
	double TTS = 0.0, ortho = inside ? 1.0 : 0.0;
	if (code == 1) {
		// safety rows of the critical samples in [0, idxHit]
#pragma unroll
		for (int k = 0; k < K; k++) {
			if (k > idxHit) { // fewer samples than rows: inert padding, :556-566
#pragma unroll
				for (int r = 0; r < NP; r++) {
					put(k * NP + r, 0.0, 1.0, -0.0);
				}
				continue;
			}
			double zk[NZ], xs[NX], h[NP], Dh[NP * NX];
			int slot = 0; // entry k of the selection, picked with selects: a dynamic index would spill the array
#pragma unroll
			for (int p = 0; p < K; p++) slot = p == k ? top.slot[p] : slot;
			const double *pp = payp(slot);
#pragma unroll
			for (int c = 0; c < NZ; c++) zk[c] = pp[c * 64];
#pragma unroll
			for (int c = 0; c < NX; c++) xs[c] = zk[c];
			M::safetySet(o, xs, h, Dh);
#pragma unroll
			for (int r = 0; r < NP; r++) {
				double Lf = 0.0, Lg = 0.0;
#pragma unroll
				for (int j = 0; j < NX; j++) {
					double s = 0.0;
#pragma unroll
					for (int c = 0; c < NX; c++) s += Dh[r + c * NP] * zk[NX + c + j * NX];
					Lf += s * f0[j];
					Lg += s * g0[j];
				}
				put(k * NP + r, Lg, h[r], -Lf);
			}
		}
		// quantities at the hitting sample (z is still the state at idxHit)
		double xh[NX], hBS, DhBS[NX], DDh[NX * NX], fCl[NX], DfCl[NX * NX];
#pragma unroll
		for (int c = 0; c < NX; c++) xh[c] = z[c];
		M::backupSet(o, xh, hBS, DhBS, DDh);
		BackupLoop<M>::closedLoop(o, xh, fCl, DfCl);
		double cosT = 0.0, n1 = 0.0, n2 = 0.0;
#pragma unroll
		for (int c = 0; c < NX; c++) {
			cosT += DhBS[c] * fCl[c];
			n1 += DhBS[c] * DhBS[c];
			n2 += fCl[c] * fCl[c];
		}
		const double den1 = sqrt(n1), den2 = sqrt(n2), den = den1 * den2;
		ortho = cosT / den; // BTorthoBS_, :516-522
		TTS = tHit;
		const double hReach = o.backTrajHorizon - tHit;
		double DhQ[NX]; // Dh_B(x_hit) Q_hit
#pragma unroll
		for (int j = 0; j < NX; j++) {
			double s = 0.0;
#pragma unroll
			for (int c = 0; c < NX; c++) s += DhBS[c] * z[NX + c + j * NX];
			DhQ[j] = s;
		}
		{ // time-to-safety row, :588-599 and :673
			double Lf = 0.0, Lg = 0.0;
#pragma unroll
			for (int j = 0; j < NX; j++) {
				const double dj = DhQ[j] / cosT;
				Lf += dj * f0[j];
				Lg += dj * g0[j];
			}
			put(K * NP, Lg, 0.0, -Lf - o.relaxTTS * hReach);
		}
		{ // orthogonality row: gradient of cos(angle(grad h_B, f_cl)) at the hit w.r.t. x0, :601-641
			double DxHit[NX * NX];
#pragma unroll
			for (int r = 0; r < NX; r++)
#pragma unroll
				for (int j = 0; j < NX; j++) DxHit[r + j * NX] = z[NX + r + j * NX] - fCl[r] * DhQ[j];
			const double den2sq = den * den;
			double Lf = 0.0, Lg = 0.0;
#pragma unroll
			for (int c = 0; c < NX; c++) {
				double Dnum = 0.0, Dd1 = 0.0, Dd2 = 0.0;
#pragma unroll
				for (int k = 0; k < NX; k++) {
					double t1 = 0.0, t2 = 0.0;
#pragma unroll
					for (int l = 0; l < NX; l++) {
						t1 += DDh[k + l * NX] * DxHit[l + c * NX];
						t2 += DfCl[k + l * NX] * DxHit[l + c * NX];
					}
					const double t3 = DhBS[k] * t2, t4 = t1 * fCl[k];
					Dd1 += t3;
					Dd2 += t4;
					Dnum += t3 + t4;
				}
				const double Dden = den2 * Dd1 / den1 + den1 * Dd2 / den2;
				const double dc = (Dnum * den - cosT * Dden) / den2sq;
				Lf += dc * f0[c];
				Lg += dc * g0[c];
			}
			put(K * NP + 1, Lg, 0.0, -Lf - o.relaxMinOrtho * (ortho - o.backTrajMinOrtho));
		}
	}
	if (a.diag) {
		a.diag[0 * ld + i] = TTS;
		a.diag[1 * ld + i] = ortho;
		a.diag[2 * ld + i] = (double)idxHit;
#pragma unroll
		for (int k = 0; k < K; k++) a.diag[(int64_t)(3 + k) * ld + i] = (code == 1 && k <= idxHit) ? (double)top.idx[k] : -1.0;
	}
	if constexpr (tb_fuse_qp<M>::value) {
	if (!a.fuseQp) { // rows out: asif_hip_assemble_batch, or a solver mode that asks for its iterations (stage 2 reads them)
		if (a.code) a.code[i] = code;
#pragma unroll
		for (int r = 0; r < NC; r++) {
			a.A[(int64_t)(r + 0 * NC) * ld + i] = rA0[r];
			a.A[(int64_t)(r + 1 * NC) * ld + i] = rA1[r];
			a.b[(int64_t)r * ld + i] = rB[r];
		}
		return;
	}
	// ---- the filter's own call, default solver mode: this lane solves its instance's QP here, with the stage that
	// decides it in stage 2 as well -- the dual active-set method, one lane per QP, rows in registers (2 x 18: 54
	// doubles) -- and stores what TbPolicy::store would store.  No rows are staged (14 + 14 MB per 32 768 segway
	// instances written and read back, and a 13 us launch, before).  An instance the stage leaves undecided (none on
	// any seeded workload) hands its rows over after all and is marked pending: stage 2 runs for the marked ones only.
	{
		QpLaneData<NV, NC> qp; // TbPolicy::load: src/asif_implicit_tb.cpp:198-210
		qp.Hd[0] = 1.0;
		qp.Hd[1] = o.relaxCost;
		qp.c[0] = -2.0 * a.udes[i];
		qp.c[1] = -2.0 * o.relaxCost * o.relaxLb;
		qp.lb[0] = o.lb[0];
		qp.lb[1] = o.relaxLb;
		qp.ub[0] = o.ub[0];
		qp.ub[1] = o.inf;
#pragma unroll
		for (int r = 0; r < NC; r++) {
			qp.A[r][0] = rA0[r];
			qp.A[r][1] = rA1[r];
			qp.b[r] = rB[r];
			qp.eq[r] = false;
		}
		// as AdmmSmall::solve with polish == 2 (admm_small.hpp): data outside the domain = the solver's max_iter verdict
		const bool nonfinite = qp_data_nonfinite<NV, NC, 1>(qp.Hd, qp.c, qp.lb, qp.ub, qp.A, qp.b);
		double sol[NV];
		int gsteps;
		// (launch_tb fuses only when no variable is pinned by its bounds, so the entry's elimination of pinned variables is
		// never taken; calling solve_general directly measured the same on the segway and 3 % slower on C12)
		const int v = GiSmall<NV, NC, 1>::solve_unchecked(qp, 0, 8 * NV + 4, sol, gsteps);
		int st = 0;
		if (nonfinite || v == kGiFailed) st = kStatusMaxIter;
		else if (v == kGiOptimal) st = kStatusSolved;
		else if (v == kGiInfeasible) st = kStatusPrimalInf;
		const bool forced = a.fuseQp == 2 || (a.fuseQp == 3 && (i & 1)); // (developer switch, see launch_tb)
		const bool pending = (st == 0 || forced) && code != -3;
		if (__any(pending)) {
			if (pending) {
#pragma unroll
				for (int r = 0; r < NC; r++) {
					a.A[(int64_t)(r + 0 * NC) * ld + i] = rA0[r];
					a.A[(int64_t)(r + 1 * NC) * ld + i] = rA1[r];
					a.b[(int64_t)r * ld + i] = rB[r];
				}
			}
		}
		a.code[i] = pending ? code + kTbPendingMark : code;
		if (!pending) { // TbPolicy::store
			if (code != -3 && st == kStatusSolved) {
				a.uact[i] = fmin(fmax(sol[0], o.lb[0]), o.ub[0]);
				a.relax[i] = sol[1];
				a.rc[i] = code; // 2 inside the backup set (:307), 1 otherwise (:343)
			} else {
				double u[1], Du[NX];
				M::backupController(o, x0, u, Du);
				a.uact[i] = fmin(fmax(u[0], o.lb[0]), o.ub[0]);
				// :315 (-1 on the trivial branch), :351 (the raw solver status leaks), :360 (-3)
				a.rc[i] = code == -3 ? ASIF_HIP_RC_BACKUP_UNREACHED : (code == 2 ? ASIF_HIP_RC_QP_FAILED : st);
			}
			if (a.diag) a.diag[(int64_t)(a.ndiag - 1) * ld + i] = 0.0; // iterations: decided before the first one
		}
	}
	}
}

template <class M>
struct TbPolicy {
	static constexpr bool kStagedRows = true; // load() reads rows from HBM (qp_kernel.hpp: XCD-contiguous blocks)
	int64_t B;
	DevOptions o;
	FilterArgs a; // a.A / a.b = staged rows, a.code = staged branch codes

	template <int NV, int NC, int G>
	__device__ __forceinline__ void load(int64_t i, int g, QpLaneData<NV, (NC + G - 1) / G> &qp) const
	{
		static_assert(NV == 2, "nu + 1 variables");
		// src/asif_implicit_tb.cpp:198-210
		qp.Hd[0] = 1.0;
		qp.Hd[1] = o.relaxCost;
		qp.c[0] = -2.0 * a.udes[i];
		qp.c[1] = -2.0 * o.relaxCost * o.relaxLb;
		qp.lb[0] = o.lb[0];
		qp.lb[1] = o.relaxLb;
		qp.ub[0] = o.ub[0];
		qp.ub[1] = o.inf;
		load_rows<NV, NC, G>(a.A, a.b, a.ld, i, g, 0ull, qp);
	}
	// fused mode (FilterArgs::fuseQp): only the instances the rows kernel marked are stage 2's
	__device__ __forceinline__ bool pending(int64_t i) const { return !a.fuseQp || a.code[i] > 4; }
	template <int NV>
	__device__ __forceinline__ void store(int64_t i, const double (&sol)[NV], int st, int it) const
	{
		int code = a.code[i];
		if (code > 4) code -= kTbPendingMark;
		if (code != -3 && st == kStatusSolved) {
			a.uact[i] = fmin(fmax(sol[0], o.lb[0]), o.ub[0]);
			a.relax[i] = sol[1];
			a.rc[i] = code; // 2 inside the backup set (:307), 1 otherwise (:343)
		} else {
			double x[M::NX], u[1], Du[M::NX];
#pragma unroll
			for (int k = 0; k < M::NX; k++) x[k] = a.x[k * a.ld + i];
			M::backupController(o, x, u, Du);
			a.uact[i] = fmin(fmax(u[0], o.lb[0]), o.ub[0]);
			// :315 (-1 on the trivial branch), :351 (the raw solver status leaks), :360 (-3)
			a.rc[i] = code == -3 ? ASIF_HIP_RC_BACKUP_UNREACHED : (code == 2 ? ASIF_HIP_RC_QP_FAILED : st);
		}
		if (a.diag) a.diag[(int64_t)(a.ndiag - 1) * a.ld + i] = (double)it;
	}
};

template <class M>
static int launch_tb(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a_in, bool assemble_only,
                     hipStream_t stream)
{
	static_assert(M::NPBTSS * M::NPSS + 2 == 18 && M::NU == 1, "QP shape 2 x 18");
	if (a_in.B <= 0) return 0;
	FilterArgs a = a_in;
	// the default solver mode decides every QP with the dual active-set stage before anything else: the rows kernel runs
	// that stage itself (see its last block) and stage 2 is left with what it marks
	// (an input pinned by its bounds, lb == ub, is eliminated by the solver's entry before the method runs: stage 2's business)
	a.fuseQp = (tb_fuse_qp<M>::value && !assemble_only && S.polish == 2 && S.lanes_per_qp == 0 && o.lb[0] != o.ub[0] &&
	            o.relaxLb < o.inf) ? 1 : 0;
	if (a.fuseQp) {
		// developer switch (tests/test_gpu_tb.py): 0 = two launches as before; 2 = the rows kernel marks EVERY instance that
		// has a QP pending, 3 = every second one -- the hand-over to stage 2 that no seeded instance takes by itself
		const char *v = getenv("ASIF_HIP_TB_FUSE");
		if (v && v[0] == '0') a.fuseQp = 0;
		else if (v && (v[0] == '2' || v[0] == '3')) a.fuseQp = v[0] - '0';
	}
	{
		constexpr size_t region = sizeof(double) * M::NPBTSS * (M::NX + M::NX * M::NX) * 64;
		const int grid = grid_for(a.B, 1, 64);
		// a region of their own for the checkpoints costs LDS-limited occupancy (two waves per CU for the segway): taken
		// when the batch does not need more than that
		const int cus = device_cus();
		if (o.integrator == 1) { // dopri5 dense output (USE_ODEINT): one pass, payload region only
			auto kern = tb_rows_kernel<M, kCkptSpill, false, true>;
			int nw = waves_per_workgroup(grid);
			while (nw > 1 && ((size_t)nw * region > 160 * 1024 || nw > tb_max_wg_waves<M>())) nw /= 2;
			if (nw * region > 48 * 1024) {
				const hipError_t he = allow_dynamic_lds((const void *)kern, nw * region);
				if (he != hipSuccess) return (int)he;
			}
			hipLaunchKernelGGL(kern, dim3((grid + nw - 1) / nw), dim3(64 * nw), nw * region, stream, o, a);
		} else {
		const int per_cu = (int)((160 * 1024) / (2 * region));
		const bool both = per_cu >= 1 && (int64_t)grid <= (int64_t)per_cu * cus;
		hipError_t he = hipSuccess;
		bool split = false;
		if constexpr (tb_split_roles<M>::value) {
			// two waves per 64 instances: only while the second wave has a SIMD to itself (two workgroups per CU)
			static const bool off = []() {
				const char *v = getenv("ASIF_HIP_TB_SPLIT"); // developer switch: 0 keeps the fused pass
				return v && v[0] == '0';
			}();
			split = !off && o.satFastOk && (int64_t)grid <= 2LL * cus && 2 * TbSplitLds<M>::bytes() <= 160 * 1024;
			if (split) {
				auto kern = tb_rows_kernel<M, kCkptSpill, true>;
				const size_t bytes = TbSplitLds<M>::bytes();
				if (bytes > 48 * 1024)
					he = allow_dynamic_lds((const void *)kern, bytes);
				if (he != hipSuccess) return (int)he;
				hipLaunchKernelGGL(kern, dim3(grid), dim3(128), bytes, stream, o, a);
			}
		}
		if (split) {
		} else if (both) {
			auto kern = tb_rows_kernel<M, kCkptLds2>;
			int nw = waves_per_workgroup(grid); // four waves per workgroup when the launch is large, as far as LDS allows
			while (nw > 1 && ((size_t)nw * 2 * region > 160 * 1024 || nw > tb_max_wg_waves<M>())) nw /= 2;
			if (nw * 2 * region > 48 * 1024)
				he = allow_dynamic_lds((const void *)kern, nw * 2 * region);
			if (he != hipSuccess) return (int)he;
			hipLaunchKernelGGL(kern, dim3((grid + nw - 1) / nw), dim3(64 * nw), nw * 2 * region, stream, o, a);
		} else {
			auto kern = tb_rows_kernel<M, kCkptSpill>;
			int nw = waves_per_workgroup(grid);
			while (nw > 1 && ((size_t)nw * region > 160 * 1024 || nw > tb_max_wg_waves<M>())) nw /= 2;
			if (nw * region > 48 * 1024)
				he = allow_dynamic_lds((const void *)kern, nw * region);
			if (he != hipSuccess) return (int)he;
			hipLaunchKernelGGL(kern, dim3((grid + nw - 1) / nw), dim3(64 * nw), nw * region, stream, o, a);
		}
		}
	}
	int e = (int)hipGetLastError();
	if (e || assemble_only) return e;
	const TbPolicy<M> p = {a.B, o, a};
	switch (S.lanes_per_qp) {
	// default 4 lanes per QP: 5 rows per lane keep the solver's state in registers (2 lanes: 9 rows per lane, 512
	// VGPR + AGPR and 144 B of spills in the ADMM / finish code); 3 us slower per 32 768 instances, measured
	case 0:
	case 4: return launch_policy<2, 18, 4>(S, p, stream);
	case 2: return launch_policy<2, 18, 2>(S, p, stream);
	case 1: return launch_policy<2, 18, 1>(S, p, stream);
	default: return ASIF_HIP_EINVAL;
	}
}

int launch_tb_segway(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                     hipStream_t stream)
{
	return launch_tb<Segway>(o, S, a, assemble_only, stream);
}

int launch_tb_pendulum(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream)
{
	return launch_tb<InvertedPendulumTB>(o, S, a, assemble_only, stream);
}

int launch_tb_di(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only, hipStream_t stream)
{
	return launch_tb<DoubleIntegratorTB>(o, S, a, assemble_only, stream);
}

} // namespace asif
