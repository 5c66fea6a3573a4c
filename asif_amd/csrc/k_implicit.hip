// k_implicit.hip -- implicit (backup-trajectory) filter: ASIFimplicit::filter, src/asif_implicit.cpp:305-356.
//
// Stage 1  implicit_rows_kernel: ASIFimplicit::updateConstraints (:403-651), one instance per lane.
//   Forward-Euler integration of the backup closed loop with its sensitivity Q = d phi_t / d x0
//   (:461-484), the safety margin sampled at every point, the npBTSS most critical samples kept by a
//   running selection (their states parked in LDS, lane-contiguous -> bank-conflict free) instead of
//   the reference's stored trajectory + full sort (:441-443,487), then the rows
//     safe row  (k,i): [ Dh_i(x_k) Q_k g | h_i(x_k) | 0 ],  b = -Dh_i(x_k) Q_k f      (:517-540,564-611)
//     backup row     : [ Dh_B(x_T) Q_T g | 0 | h_B(x_T) ],  b = -Dh_B(x_T) Q_T f      (:542-554)
//   written SoA to HBM (the same layout asif_hip_assemble_batch hands out).
// Stage 2  qp_policy_kernel<3,41,G>: cost/bounds of initialize()/updateCost() (:237-254,653-664),
//   in-register ADMM, then the epilogue of filter(): inputSaturate + relax on success (:338-347),
//   saturated backup controller and rc = -1 on failure (:348-355).
// ASIFimplicitRB (src/asif_implicit_robust.cpp, RB = true): the same two stages with
//   * the backup input held over backContDt along the trajectory (:891-903),
//   * the safe rows' margins replaced by the lower end of the interval safety set over x_k +- x_unc (:635-647),
//   * optionally two small ReLU networks evaluated on [x; Dh_index_] whose outputs are added to the first
//     row's Lfh / Lgh (include/asif_learning_utils.h:123-155); Dh_index_ is the first column of Dh_SS Q at the
//     most critical sample, or at sample n_debug (:590-605,624-632).
//   The interval Lie derivatives the reference also forms (:698-709) feed nothing there and are not computed.
// The 5000-step trajectory is ~99 % of the work and is inherently sequential per instance; the solve
// is re-dealt G lanes per QP so the second launch fills the whole chip.
#include <type_traits>
#include "backup_traj.hpp"
#include <utility>
#include "qp_kernel.hpp"

namespace asif {

constexpr int kLearnMaxHidden = 32; // widest hidden layer the rows kernel keeps in LDS (2 x 32 x 64 doubles)

// driftNN / actNN (include/asif_learning_utils.h:34-121), one instance per lane: weights are wave-uniform
// (scalar loads), the two hidden activations live in LDS as [unit][lane].  The reference zero-pads the input
// to d_*_in; the padded columns multiply zeros and are skipped.
template <int NIN>
__device__ inline double learn_net(const DevOptions::Learn &L, int net, const double (&in)[NIN], double *act,
                                         int lane, int outIdx)
{
	const int h1 = L.dHidden[net], h2 = L.dHidden2[net];
	const double *__restrict__ w1 = L.w1[net], *__restrict__ b1 = L.b1[net];
	const double *__restrict__ w2 = L.w2[net], *__restrict__ b2 = L.b2[net];
	const double *__restrict__ w3 = L.w3[net], *__restrict__ b3 = L.b3[net];
	double *a1 = act, *a2 = act + kLearnMaxHidden * 64;
	for (int j = 0; j < h1; j++) {
		double s = 0.0;
#pragma unroll
		for (int i = 0; i < NIN; i++) s += w1[j + i * h1] * in[i];
		a1[j * 64 + lane] = fmax(0., s + b1[j]);
	}
	for (int j = 0; j < h2; j++) {
		double s = 0.0;
		for (int i = 0; i < h1; i++) s += w2[j + i * h2] * a1[i * 64 + lane];
		a2[j * 64 + lane] = fmax(0., s + b2[j]);
	}
	const int nout = L.dOut[net];
	double s = 0.0;
	for (int i = 0; i < h2; i++) s += w3[outIdx + i * nout] * a2[i * 64 + lane];
	return s + b3[outIdx];
}

// DOPRI: the backup trajectory of the reference's USE_ODEINT build (asif_hip_options::integrator = 1): adaptive
// dopri5 with dense output at the sample times instead of forward Euler.  The adaptive step straddles samples, so
// there is no per-block restart point: one pass, the exact per-sample selection with the states parked in LDS
// (the handful of Runge-Kutta steps per trajectory is cheap next to 5000 Euler steps; the samples are interpolations).

// CKPT: where the block-start states of pass 1's running selection live (K slots of NZ + 2 doubles per lane), as in
// k_tb.hip: kCkptLds2 -- in LDS next to pass 2's payload, nothing of the search touches HBM (70 KB per wave for the
// pendulum: two waves per CU, taken when the batch needs no more: C3 is 256 waves); kCkptSpill -- in LDS during pass 1,
// in the region pass 2 reuses, the K survivors written to HBM once between the passes.  (Round 2 wrote a checkpoint
// whenever a block entered the selection.)
constexpr int kImCkptLds2 = 1, kImCkptSpill = 2;
template <class M>
struct ImplicitLds {
	static constexpr int NZ = M::NX + M::NX * M::NX, K = M::NPBTSS, CK = NZ + 2;
	static constexpr int kPay = K * NZ * 64, kCk = K * CK * 64;
	static constexpr size_t bytes(int ckpt, bool rb)
	{
		const size_t head = ckpt == kImCkptLds2 ? (size_t)kPay + kCk : (size_t)(kPay > kCk ? kPay : kCk);
		return sizeof(double) * (head + (rb ? 2 * kLearnMaxHidden * 64 : 0));
	}
};

// models that declare kImFuseQp have the rows kernel of plain ASIFimplicit solve the instance's QP itself (the kernel's
// last block): DoubleIntegrator_implicit's 3 x 17 fits registers (51 doubles), the pendulum's 3 x 41 does not
template <class M, class = void>
struct im_fuse_qp : std::false_type {};
template <class M>
struct im_fuse_qp<M, std::enable_if_t<M::kImFuseQp>> : std::true_type {};
// a.code of an instance whose QP the rows kernel left to stage 2 (fused mode): 1 + this
constexpr int kImPendingMark = 8;
template <int... I, class F>
__device__ __forceinline__ void unroll_seq(std::integer_sequence<int, I...>, F &&f)
{
	(f(std::integral_constant<int, I>()), ...);
}

template <class M, bool RB, bool DOPRI = false, int CKPT = kImCkptSpill>
__global__ __launch_bounds__(256) void implicit_rows_kernel(DevOptions o_arg, FilterArgs a)
{
	// the soft saturation selects between these two and the input: as kernel arguments (SGPRs) they are copied into
	// VGPRs at every Euler step; an opaque copy made once, here, stays in two VGPR pairs for the whole kernel
	DevOptions o = o_arg;
	asm("" : "+v"(o.lb[0]), "+v"(o.ub[0]));
	static_assert(!(RB && DOPRI), "the held input of ASIFimplicitRB makes the rhs time-dependent: Euler only");
	constexpr int NX = M::NX, NP = M::NPSS, K = M::NPBTSS, NB = M::NPBS, NZ = NX + NX * NX;
	constexpr int NC = K * NP + NB;
	static_assert(NB == 1, "one backup-set function");
	static_assert(!RB || NP >= NX, "Dh_index_ takes the first nx entries of a column of the npSS x nx product");
	extern __shared__ double im_lds[];
	using L = ImplicitLds<M>;
	// a workgroup is one to four waves that share nothing (launchers.hpp: waves_per_workgroup): each has its own regions
	const int wv = (int)(threadIdx.x >> 6);
	double *const wlds = im_lds + (size_t)wv * (L::bytes(CKPT, RB) / sizeof(double));
	double *const pay = wlds;                                                 // pass 2: states of the K most critical samples
	double *const ckl = CKPT == kImCkptLds2 ? wlds + L::kPay : wlds;          // pass 1: checkpoints of the K selected blocks
	double *const act = wlds + (CKPT == kImCkptLds2 ? L::kPay + L::kCk : (L::kPay > L::kCk ? L::kPay : L::kCk));
	(void)act;
	(void)ckl;
	const int lane = (int)(threadIdx.x & 63);
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i - lane >= a.B) return; // a wave past the end of the batch (wave-uniform; the kernel has no barrier)
	const bool live = i < a.B;
	if (!live) i = a.B - 1;

	double x0[NX], f0[NX], g0[NX];
#pragma unroll
	for (int k = 0; k < NX; k++) x0[k] = a.x[k * a.ld + i];
	M::dynamics(o, x0, f0, g0); // :414

	double z[NZ];
#pragma unroll
	for (int k = 0; k < NZ; k++) z[k] = 0.0;
#pragma unroll
	for (int k = 0; k < NX; k++) {
		z[k] = x0[k];
		z[NX + k * (NX + 1)] = 1.0; // Q(0) = I, :417-425
	}
	typedef typename BackupLoop<M>::Hold Hold;
	Hold hold = {0.0, 0.0};
	double zDbg[NZ]; // the sample feeding the networks when n_debug selects one (:590-605)
#pragma unroll
	for (int k = 0; k < NZ; k++) zDbg[k] = z[k];
	TopK<K> top;
	top.init();
	double zEnd[NZ];
	if constexpr (DOPRI) {
		Dopri5<M> rk;
		rk.init(o, z, o.trajDt);
		int guard = 200000; // odeint gives up after 500 failed attempts of one step; a stuck controller must not hang a wave
#pragma unroll 1
		for (int s = 0; s < o.npBT; s++) {
			const double ts = o.trajDt * (double)s; // backTraj_[i].first, :451
			// n_step_iterator: step while t < ts (less_with_sign: by more than epsilon), then interpolate
			while (guard > 0 && __any(rk.behind(ts))) {
				rk.tryStep(o, rk.behind(ts));
				guard--;
			}
			rk.failed = rk.failed | rk.behind(ts); // the wave's budget ran out with this lane still short of ts
			double zs[NZ], xs[NX];
			rk.dense(ts, zs);
#pragma unroll
			for (int k = 0; k < NX; k++) xs[k] = zs[k];
			const double hm = M::safetyMin(o, xs);
			if (__any(hm < top.key[K - 1])) {
				const int slot = top.insert(hm, s);
				if (slot >= 0) {
#pragma unroll
					for (int k = 0; k < NZ; k++) pay[(slot * NZ + k) * 64 + lane] = zs[k];
				}
			}
#pragma unroll
			for (int k = 0; k < NZ; k++) zEnd[k] = zs[k];
		}
	} else {
	// ---- pass 1: the whole trajectory, nothing kept per sample.  Per block of MB consecutive samples the smallest
	// margin of the block feeds a running selection of the K blocks with the smallest minima (ties -> earlier
	// block).  Every one of the K most critical SAMPLES (value, then index) lies in one of those K blocks: a block
	// outside them is preceded by K blocks whose first minimum is a distinct sample ordered before every sample of it.
	// The state at a block's first sample is held in registers over the block and goes to HBM only when the block
	// enters the selection, into the slot of the entry it displaces: a.ckpt is [K][CK][ld] -- K checkpoints per
	// instance (640 B for the pendulum) instead of one per block (20 KB), written a handful of times per trajectory
	// instead of once per block.
	// The per-sample selection network this replaces ran on nearly every step (some lane of the wave inserts)
	// and cost 35 % of the kernel.
	constexpr int MB = M::kTrajBlock, CK = NZ + 2;
	TopK<K> topB;
	topB.init();
	RunningMargin<M> brun; // smallest margin of the current block
	brun.reset();
	double *ck = a.ckpt + i;
	const int64_t ldc = a.ld;
	double zs[NZ]; // state (and hold) at the first sample of the current block
	Hold hs = hold;
#pragma unroll
	for (int k = 0; k < NZ; k++) zs[k] = z[k];
	auto commit = [&](int blk) { // close block blk: keep its checkpoint if it is among the K most critical so far
		const double bmin = brun.value();
		if (__any(bmin < topB.key[K - 1])) {
			const int slot = topB.insert(bmin, blk);
			if (slot >= 0) {
				double *c = ckl + slot * CK * 64 + lane;
#pragma unroll
				for (int k = 0; k < NZ; k++) c[k * 64] = zs[k];
				c[NZ * 64] = hs.u;
				c[(NZ + 1) * 64] = hs.tLast;
			}
		}
	};
	// One block at a time: its samples, then the step into the next block's first sample.  The block runs on the
	// branch-free fast path of the model's sin / cos.  Its argument range is policed per block, not per step: either the
	// fast path poisons the state of a lane whose argument left the range (kTrigPoison), or -- where the safety margin
	// this loop tracks anyway bounds the argument (kTrigBoundedByMargin) -- the block's smallest margin is looked at
	// (kTrigUnchecked: every step of run() starts from a sample whose margin went into bmin).  Either way the block is
	// repeated from its start state with the checking version: same results as checking at every step, without a
	// wave-level branch (and, by margin, without any per-step work) in the common step.
	// The fast path also takes the soft saturation's short forms (BackupLoop::saturateSoft<FAST>), valid for ordinary
	// saturation constants (DevOptions::satFastOk, checked on the host); other options run the generic step.
	// the fast step's trig mode: carried along the block, bounded by the margin, or poisoning (models.hpp)
	constexpr int kFastTrig = trig_carry<M>::value ? kTrigCarried : (trig_by_margin<M>::value ? kTrigUnchecked : kTrigPoison);
	TrigCarry carry = {0.0, 0.0, 1.0};
	const int nblk = (o.npBT + MB - 1) / MB;
	bool anyRedo = false; // some block of this wave ran on the generic step (pass 2 then does too)
#pragma unroll 1
	for (int blk = 0; blk < nblk; blk++) {
		const int s0 = blk * MB;
		const int n = (o.npBT - s0) < MB ? (o.npBT - s0) : MB; // samples of this block
		const bool more = s0 + n < o.npBT;
		bool bevelSeen = false; // bevel-free flavour: some step of the block met a bevel after all
		auto run = [&](auto fast, auto nobevel) {
			constexpr int P = !decltype(fast)::value ? kTrigChecked : kFastTrig;
			constexpr bool NB = decltype(nobevel)::value;
			brun.reset();
			// reset: the step out of the block's first sample evaluates sin / cos afresh (kTrigCarried); a literal at every
			// call site, so that the step is compiled in its two forms instead of choosing at run time
			auto sample = [&](int k, bool reset) {
				const int sidx = s0 + k;
				if (k > 0) BackupLoop<M>::template eulerStepT<RB, P, NB>(o, z, hold, (double)(unsigned)sidx * o.trajDt, &carry, reset, &bevelSeen);
				if (RB && sidx == o.nDebug) {
#pragma unroll
					for (int c = 0; c < NZ; c++) zDbg[c] = z[c];
				}
				double xs[NX];
#pragma unroll
				for (int c = 0; c < NX; c++) xs[c] = z[c];
				brun.add(o, xs);
			};
			static_assert(MB >= 4 && MB % 4 == 0, "blocks of a multiple of four samples");
			if (n == MB) { // full block: compile-time trip count, unrolled by four (loop control is SALU + a branch per step)
				if constexpr (trig_carry<M>::value) {
					sample(0, false);
					sample(1, true);
					sample(2, false);
					sample(3, false);
					if constexpr (!RB) { // the whole block in line: no loop control at all (C3 776 -> 763 us; the RB step is
						               // larger and loses 2 % this way)
#pragma unroll
						for (int k = 4; k < MB; k++) sample(k, false);
					} else {
#pragma unroll 4
						for (int k = 4; k < MB; k++) sample(k, false);
					}
				} else {
#pragma unroll 4
					for (int k = 0; k < MB; k++) sample(k, false);
				}
			} else {
				sample(0, false);
				if (n > 1) sample(1, true);
#pragma unroll 1
				for (int k = 2; k < n; k++) sample(k, false);
			}
			if (more) { // n == MB here (only the last block is partial)
				BackupLoop<M>::template eulerStepT<RB, P, NB>(o, z, hold, (double)(unsigned)(s0 + n) * o.trajDt, &carry, false, &bevelSeen);
			}
		};
		bool redo = !o.satFastOk; // options outside the fast step's preconditions (uniform): generic step throughout
		// A block none of whose lanes starts near a bevel runs the step without the bevel's divergent block and its
		// branch (a branch costs a lone wave ~45 cycles, taken or not: an eighth of this step).  "Near" is a prediction
		// (bevel_rate, models.hpp); bevelSeen is the check, and a block that met a bevel anyway is repeated.
		bool far = false;
		if constexpr (bevel_rate<M>::value > 0.0) {
			if (!redo && n == MB && o.bevelFree) {
				double xs[NX], u0[1], Du0[NX];
#pragma unroll
				for (int c = 0; c < NX; c++) xs[c] = z[c];
				M::backupController(o, xs, u0, Du0);
				const double au = fabs(((RB && o.backContDt > 0 ? hold.u : u0[0]) - o.satMiddle) * o.twoOverRange);
				const double d = o.bevelFree == 2 ? 0.0 : bevel_rate<M>::value * (double)MB * o.trajDt;
				far = !__any(au > o.bevelStart - d && au < o.bevelStop + d);
			}
		}
		if (far) {
			bevelSeen = false;
			run(std::true_type(), std::true_type());
			bool bad = bevelSeen;
#pragma unroll
			for (int c = 0; c < NZ; c++) bad = bad || (c < NX ? !(fabs(z[c]) < kStateSane) : (z[c] != z[c]));
			if constexpr (trig_carry<M>::value) bad = bad || !M::trigCarryBounded(o, brun.value());
			else if constexpr (trig_by_margin<M>::value) bad = bad || !M::trigArgsBounded(brun.value());
			far = !__any(bad);
			if (!far) { // mispredicted (or an alarm): from the block's start again, with the full fast step first
#pragma unroll
				for (int c = 0; c < NZ; c++) z[c] = zs[c];
				hold = hs;
			}
		}
		if (!redo && !far) {
			run(std::true_type(), std::false_type());
			bool bad = false;
#pragma unroll
			for (int c = 0; c < NZ; c++) bad = bad || (c < NX ? !(fabs(z[c]) < kStateSane) : (z[c] != z[c])); // x: NaN or beyond any sane magnitude; Q: NaN (a stiff model's sensitivity may overflow under forward Euler, as it does upstream)
			if constexpr (trig_carry<M>::value) bad = bad || !M::trigCarryBounded(o, brun.value());
			else if constexpr (trig_by_margin<M>::value) bad = bad || !M::trigArgsBounded(brun.value());
			redo = __any(bad); // never on sane trajectories
			if (redo) {
#pragma unroll
				for (int c = 0; c < NZ; c++) z[c] = zs[c];
				hold = hs;
			}
		}
		if (redo) {
			run(std::false_type(), std::false_type());
			anyRedo = true;
		}
		commit(blk);
#pragma unroll
		for (int c = 0; c < NZ; c++) zs[c] = z[c];
		hs = hold;
	}
	// z now holds the state of the last sample: the loop's last block has no step beyond it
#pragma unroll
	for (int k = 0; k < NZ; k++) zEnd[k] = z[k];

	if constexpr (CKPT == kImCkptSpill) {
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
				double *c = ck + (int64_t)slot * CK * ldc;
				const double *l = ckl + slot * CK * 64 + lane;
#pragma unroll
				for (int k = 0; k < CK; k++) c[k * ldc] = l[k * 64];
			}
		}
	}
	// ---- pass 2: re-integrate the selected blocks from their checkpoints, in increasing block order so that
	// samples arrive in increasing index (the selection's tie rule: earlier sample first), and run the exact
	// per-sample selection with the states parked in LDS -- at most K*MB of the npBT steps.
	int cur = -1;
	const bool fast2 = o.satFastOk && !anyRedo; // wave-uniform
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
		if constexpr (CKPT == kImCkptSpill) {
			const double *c = ck + (int64_t)(have ? sl : 0) * CK * ldc;
#pragma unroll
			for (int k = 0; k < NZ; k++) z[k] = c[k * ldc];
			hold.u = c[NZ * ldc];
			hold.tLast = c[(NZ + 1) * ldc];
		} else {
			const double *c = ckl + (have ? sl : 0) * CK * 64 + lane;
#pragma unroll
			for (int k = 0; k < NZ; k++) z[k] = c[k * 64];
			hold.u = c[NZ * 64];
			hold.tLast = c[(NZ + 1) * 64];
		}
#pragma unroll 1
		for (int t = 0; t < MB; t++) {
			const int s = blk * MB + t;
			if (t > 0) {
				// the samples of a selected block went through pass 1's range check with these very states: the fast step
				// is valid for them again (steps past the horizon or of a lane without a block feed nothing)
				if (fast2) {
					if (t == 1) BackupLoop<M>::template eulerStepT<RB, kFastTrig>(o, z, hold, (double)(unsigned)s * o.trajDt, &carry, true);
					else BackupLoop<M>::template eulerStepT<RB, kFastTrig>(o, z, hold, (double)(unsigned)s * o.trajDt, &carry, false);
				}
				else BackupLoop<M>::template eulerStepT<RB>(o, z, hold, (double)(unsigned)s * o.trajDt);
			}
			double xs[NX];
#pragma unroll
			for (int k = 0; k < NX; k++) xs[k] = z[k];
			const double hm = (have && s < o.npBT) ? M::safetyMin(o, xs) : __builtin_huge_val();
			if (__any(hm < top.key[K - 1])) {
				const int slot = top.insert(hm, s);
				if (slot >= 0) {
#pragma unroll
					for (int k = 0; k < NZ; k++) pay[(slot * NZ + k) * 64 + lane] = z[k];
				}
			}
		}
	}
	} // Euler
#pragma unroll
	for (int k = 0; k < NZ; k++) z[k] = zEnd[k];
	if (!live) return;
	const int64_t ld = a.ld;
	double dhIndex[NX], Lf00 = 0.0, Lg00 = 0.0; // Dh_index_[0..nx), Lfh[0], Lgh[0]
#pragma unroll
	for (int c = 0; c < NX; c++) dhIndex[c] = 0.0;
	// Models that declare kImFuseQp keep the rows of plain ASIFimplicit (the ASIFimplicitRB instantiation with the solve
	// in it measured 2.4 % slower on C10: its loop carries the held input and the interval margins as well) -- keep the rows of the instance -- [Lgh_r, h_r | -Lfh_r],
	// three doubles each -- in registers until the end of the kernel and solve the QP right here (last block); the others
	// write each row out as it is made.  `row` is a compile-time value at every call of put().
	constexpr bool kHold = im_fuse_qp<M>::value && !RB;
	double rLg[kHold ? NC : 1], rH[kHold ? NC : 1], rB[kHold ? NC : 1];
	auto put = [&](int row, double lg, double hv, double bb, bool last) {
		if constexpr (kHold) {
			rLg[row] = lg;
			rH[row] = hv;
			rB[row] = bb;
		} else {
			a.A[(int64_t)(row + 0 * NC) * ld + i] = lg;
			if (a.compactRows) { // (wave-uniform) the row's one margin entry; the loader knows its column
				a.A[(int64_t)(row + 1 * NC) * ld + i] = hv;
			} else {
				a.A[(int64_t)(row + 1 * NC) * ld + i] = last ? 0.0 : hv;
				a.A[(int64_t)(row + 2 * NC) * ld + i] = last ? hv : 0.0;
			}
			a.b[(int64_t)row * ld + i] = bb;
		}
	};
	// safe rows from the parked critical samples
	auto sample_rows = [&](auto kk) {
		const int k = kk;
		double zk[NZ], xs[NX], h[NP], Dh[NP * NX];
		int slot = 0, sidx = 0; // entry k of the selection, picked with selects: a dynamic index would spill the arrays
#pragma unroll
		for (int p = 0; p < K; p++) {
			slot = p == k ? top.slot[p] : slot;
			sidx = p == k ? top.idx[p] : sidx;
		}
#pragma unroll
		for (int c = 0; c < NZ; c++) zk[c] = pay[(slot * NZ + c) * 64 + lane];
#pragma unroll
		for (int c = 0; c < NX; c++) xs[c] = zk[c];
		M::safetySet(o, xs, h, Dh);
		if (RB) M::safetySetLo(o, xs, h); // :635-647
#pragma unroll
		for (int r = 0; r < NP; r++) {
			double DhQ[NX];
#pragma unroll
			for (int j = 0; j < NX; j++) {
				double s = 0.0;
#pragma unroll
				for (int c = 0; c < NX; c++) s += Dh[r + c * NP] * zk[NX + c + j * NX];
				DhQ[j] = s;
			}
			double Lf = 0.0, Lg = 0.0;
#pragma unroll
			for (int j = 0; j < NX; j++) {
				Lf += DhQ[j] * f0[j];
				Lg += DhQ[j] * g0[j];
			}
			if (RB && k == 0) {
				if (r < NX) dhIndex[r] = DhQ[0]; // :624-632, Dh_index_[i] = (Dh_SS Q)[i + 0*npSS]
				if (r == 0) {
					Lf00 = Lf;
					Lg00 = Lg;
				}
			}
			put(k * NP + r, Lg, h[r], -Lf, false);
		}
		if (a.diag) a.diag[(int64_t)k * ld + i] = (double)sidx;
	};
	if constexpr (kHold) {
		unroll_seq(std::make_integer_sequence<int, K>(), sample_rows);
	} else {
#pragma unroll 1
		for (int k = 0; k < K; k++) sample_rows(k);
	}
	// backup-set row at the end of the trajectory
	{
		double xs[NX], hB, DhB[NX], DDh[NX * NX];
#pragma unroll
		for (int c = 0; c < NX; c++) xs[c] = z[c];
		M::backupSet(o, xs, hB, DhB, DDh);
		double Lf = 0.0, Lg = 0.0;
#pragma unroll
		for (int j = 0; j < NX; j++) {
			double s = 0.0;
#pragma unroll
			for (int c = 0; c < NX; c++) s += DhB[c] * z[NX + c + j * NX];
			Lf += s * f0[j];
			Lg += s * g0[j];
		}
		put(K * NP, Lg, hB, -Lf, true);
	}
	if (RB) {
		if (o.nDebug >= 0) { // :590-605: Dh_index_ from sample n_debug instead
			double xs[NX], h[NP], Dh[NP * NX];
#pragma unroll
			for (int c = 0; c < NX; c++) xs[c] = zDbg[c];
			M::safetySet(o, xs, h, Dh);
#pragma unroll
			for (int r = 0; r < NX; r++) {
				double s = 0.0;
#pragma unroll
				for (int c = 0; c < NX; c++) s += Dh[r + c * NP] * zDbg[NX + c + 0 * NX];
				dhIndex[r] = s;
			}
		}
		double dLf = 0.0, dLg = 0.0;
		if (o.useLearning) { // update_weights, include/asif_learning_utils.h:123-155
			double in[2 * NX];
#pragma unroll
			for (int c = 0; c < NX; c++) {
				in[c] = x0[c];
				in[NX + c] = dhIndex[c];
			}
			dLf = learn_net<2 * NX>(o.learn, 0, in, act, lane, 0);
			dLg = learn_net<2 * NX>(o.learn, 1, in, act, lane, 0);
			a.A[(int64_t)(0 + 0 * NC) * ld + i] = Lg00 + dLg;
			a.b[(int64_t)0 * ld + i] = -(Lf00 + dLf);
		}
		if (a.diag && a.ndiag >= K + NX + 3) { // public members Dh_index_, learning_data_.Lfh_diff / Lgh_diff
#pragma unroll
			for (int c = 0; c < NX; c++) a.diag[(int64_t)(K + c) * ld + i] = dhIndex[c];
			a.diag[(int64_t)(K + NX) * ld + i] = dLf;
			a.diag[(int64_t)(K + NX + 1) * ld + i] = dLg;
		}
	}
	if constexpr (!kHold) {
		if (a.code) a.code[i] = 1;
	} else {
		if (!a.fuseQp) { // rows out: asif_hip_assemble_batch (the reference's full block), or a solver mode that asks for its iterations
			if (a.code) a.code[i] = 1;
#pragma unroll
			for (int r = 0; r < NC; r++) {
				const bool last = r == NC - NB;
				a.A[(int64_t)(r + 0 * NC) * ld + i] = rLg[r];
				if (a.compactRows) {
					a.A[(int64_t)(r + 1 * NC) * ld + i] = rH[r];
				} else {
					a.A[(int64_t)(r + 1 * NC) * ld + i] = last ? 0.0 : rH[r];
					a.A[(int64_t)(r + 2 * NC) * ld + i] = last ? rH[r] : 0.0;
				}
				a.b[(int64_t)r * ld + i] = rB[r];
			}
			return;
		}
		// ---- the filter's own call, default solver mode: this lane solves its instance's QP here with the stage that
		// decides it in stage 2 as well (dual active-set method, one lane per QP, rows in registers) and stores what
		// ImplicitPolicy::store would store; nothing is staged.  An instance the stage leaves undecided (none on any
		// seeded workload) hands its rows over and is marked in `code`: stage 2 runs for the marked ones only.
		constexpr int NV = 3;
		QpLaneData<NV, NC> qp; // ImplicitPolicy::load: src/asif_implicit.cpp:237-254
		qp.Hd[0] = 1.0;
		qp.Hd[1] = o.relaxCost;
		qp.Hd[2] = o.relaxCost;
		qp.c[0] = -2.0 * a.udes[i];
		qp.c[1] = -2.0 * o.relaxCost * o.relaxLb;
		qp.c[2] = -2.0 * o.relaxCost * o.relaxReachLb;
		qp.lb[0] = o.lb[0];
		qp.lb[1] = o.relaxLb;
		qp.lb[2] = o.relaxReachLb;
		qp.ub[0] = o.ub[0];
		qp.ub[1] = o.inf;
		qp.ub[2] = o.inf;
#pragma unroll
		for (int r = 0; r < NC; r++) {
			const bool last = r == NC - NB;
			qp.A[r][0] = rLg[r];
			qp.A[r][1] = last ? 0.0 : rH[r];
			qp.A[r][2] = last ? rH[r] : 0.0;
			qp.b[r] = rB[r];
			qp.eq[r] = false;
		}
		const bool nonfinite = qp_data_nonfinite<NV, NC, 1>(qp.Hd, qp.c, qp.lb, qp.ub, qp.A, qp.b);
		double sol[NV];
		int gsteps;
		// (solve_general: the launcher fuses only when no variable is pinned by its bounds -- fuse_mode -- so the entry's
		// elimination of pinned variables, which would hold a second copy of the rows, has nothing to do here)
		const int v = GiSmall<NV, NC, 1>::solve_general(qp, 0, 8 * NV + 4, sol, gsteps);
		int st = 0; // as AdmmSmall::solve with polish == 2 (admm_small.hpp)
		if (nonfinite || v == kGiFailed) st = kStatusMaxIter;
		else if (v == kGiOptimal) st = kStatusSolved;
		else if (v == kGiInfeasible) st = kStatusPrimalInf;
		const bool forced = a.fuseQp == 2 || (a.fuseQp == 3 && (i & 1)); // (developer switch, see launch_implicit_di)
		const bool pending = st == 0 || forced;
		if (__any(pending)) {
			if (pending) {
#pragma unroll
				for (int r = 0; r < NC; r++) { // compact rows, as stage 2's loader reads them
					a.A[(int64_t)(r + 0 * NC) * ld + i] = rLg[r];
					a.A[(int64_t)(r + 1 * NC) * ld + i] = rH[r];
					a.b[(int64_t)r * ld + i] = rB[r];
				}
			}
		}
		a.code[i] = pending ? 1 + kImPendingMark : 1;
		if (!pending) { // ImplicitPolicy::store
			if (st == kStatusSolved) {
				a.uact[i] = fmin(fmax(sol[0], o.lb[0]), o.ub[0]);
				a.relax[i] = sol[1];
				a.relax[ld + i] = sol[2];
				a.rc[i] = ASIF_HIP_RC_OK;
			} else {
				double u[1], Du[NX];
				M::backupController(o, x0, u, Du);
				a.uact[i] = fmin(fmax(u[0], o.lb[0]), o.ub[0]);
				a.rc[i] = ASIF_HIP_RC_QP_FAILED;
			}
			if (a.diag) a.diag[(int64_t)(a.ndiag - 1) * ld + i] = 0.0; // iterations: decided before the first one
		}
	}
}

template <class M>
struct ImplicitPolicy {
	static constexpr bool kStagedRows = true; // load() reads rows from HBM (qp_kernel.hpp: XCD-contiguous blocks)
	int64_t B;
	DevOptions o;
	FilterArgs a; // a.A / a.b = staged rows

	template <int NV, int NC, int G>
	__device__ __forceinline__ void load(int64_t i, int g, QpLaneData<NV, (NC + G - 1) / G> &qp) const
	{
		static_assert(NV == 3, "nu + 2 variables");
		qp.Hd[0] = 1.0;
		qp.Hd[1] = o.relaxCost;
		qp.Hd[2] = o.relaxCost;
		qp.c[0] = -2.0 * a.udes[i];
		qp.c[1] = -2.0 * o.relaxCost * o.relaxLb;
		qp.c[2] = -2.0 * o.relaxCost * o.relaxReachLb;
		qp.lb[0] = o.lb[0];
		qp.lb[1] = o.relaxLb;
		qp.lb[2] = o.relaxReachLb;
		qp.ub[0] = o.ub[0];
		qp.ub[1] = o.inf;
		qp.ub[2] = o.inf;
		// compact rows (FilterArgs::compactRows): [Lgh_r, h_r | -Lfh_r]; h_r multiplies the safe relaxation on the
		// safety rows and the reach relaxation on the last row, the backup set's (src/asif_implicit.cpp:591-611)
		constexpr int RPL = (NC + G - 1) / G;
#pragma unroll
		for (int k = 0; k < RPL; k++) {
			const int r = g + k * G;
			const bool valid = r < NC;
			const int rr = valid ? r : 0;
			const double lg = a.A[(int64_t)rr * a.ld + i], hv = a.A[(int64_t)(rr + NC) * a.ld + i];
			const double bv = a.b[(int64_t)rr * a.ld + i];
			const bool last = rr == NC - M::NPBS;
			static_assert(M::NPBS == 1, "one backup-set row");
			qp.A[k][0] = valid ? lg : 0.0;
			qp.A[k][1] = (valid && !last) ? hv : 0.0;
			qp.A[k][2] = (valid && last) ? hv : 0.0;
			qp.b[k] = valid ? bv : -1e20; // out-of-range rows are inert: 0.x >= -big
			qp.eq[k] = false;
		}
	}
	// fused mode (FilterArgs::fuseQp): only the instances the rows kernel marked are stage 2's
	__device__ __forceinline__ bool pending(int64_t i) const { return !a.fuseQp || a.code[i] > kImPendingMark; }
	template <int NV>
	__device__ __forceinline__ void store(int64_t i, const double (&sol)[NV], int st, int it) const
	{
		if (st == kStatusSolved) {
			a.uact[i] = fmin(fmax(sol[0], o.lb[0]), o.ub[0]);
			a.relax[i] = sol[1];
			a.relax[a.ld + i] = sol[2];
			a.rc[i] = ASIF_HIP_RC_OK;
		} else {
			double x[M::NX], u[1], Du[M::NX];
#pragma unroll
			for (int k = 0; k < M::NX; k++) x[k] = a.x[k * a.ld + i];
			M::backupController(o, x, u, Du);
			a.uact[i] = fmin(fmax(u[0], o.lb[0]), o.ub[0]);
			a.rc[i] = ASIF_HIP_RC_QP_FAILED;
		}
		if (a.diag) a.diag[(int64_t)(a.ndiag - 1) * a.ld + i] = (double)it;
	}
};

template <class K>
static int launch_with_lds(K kern, size_t bytes, int64_t waves, int nw, const DevOptions &o, const FilterArgs &a, hipStream_t stream)
{
	bytes *= (size_t)nw;
	if (bytes > 48 * 1024) {
		const hipError_t he = allow_dynamic_lds((const void *)kern, bytes);
		if (he != hipSuccess) return (int)he;
	}
	hipLaunchKernelGGL(kern, dim3((unsigned)((waves + nw - 1) / nw)), dim3(64 * nw), bytes, stream, o, a);
	return (int)hipGetLastError();
}

// rows kernel of model M: checkpoints in an LDS region of their own when the batch fits the occupancy that leaves
template <class M>
static int launch_rows(const DevOptions &o, const FilterArgs &a, hipStream_t stream, bool rb)
{
	using L = ImplicitLds<M>;
	const int64_t waves = grid_for(a.B, 1, 64);
	const int cus = device_cus();
	constexpr size_t kLds = 160 * 1024;
	// four waves per workgroup when the launch is large (waves_per_workgroup), as far as their regions fit a CU's LDS
	auto nwaves = [&](size_t bytes) {
		int nw = waves_per_workgroup(waves);
		while (nw > 1 && (size_t)nw * bytes > kLds) nw /= 2;
		return nw;
	};
	if (o.integrator == 1) { // one pass, no checkpoints: payload region only
		const size_t b = L::bytes(kImCkptSpill, false);
		return launch_with_lds(implicit_rows_kernel<M, false, true>, b, waves, nwaves(b), o, a, stream);
	}
	const size_t both = L::bytes(kImCkptLds2, rb), one = L::bytes(kImCkptSpill, rb);
	const int per_cu = (int)(kLds / both);
	const bool lds2 = per_cu >= 1 && waves <= (int64_t)(per_cu < 4 ? per_cu : 4) * cus;
	const size_t b = lds2 ? both : one;
	const int nw = nwaves(b);
	if (rb) {
		if (lds2) return launch_with_lds(implicit_rows_kernel<M, true, false, kImCkptLds2>, b, waves, nw, o, a, stream);
		return launch_with_lds(implicit_rows_kernel<M, true, false, kImCkptSpill>, b, waves, nw, o, a, stream);
	}
	if (lds2) return launch_with_lds(implicit_rows_kernel<M, false, false, kImCkptLds2>, b, waves, nw, o, a, stream);
	return launch_with_lds(implicit_rows_kernel<M, false, false, kImCkptSpill>, b, waves, nw, o, a, stream);
}

// FilterArgs::fuseQp of a filter call: the default solver mode decides every QP with the dual active-set stage before
// anything else, so the rows kernel of plain ASIFimplicit runs that stage itself (models with kImFuseQp) and stage 2 is
// left with what it marks.  ASIF_HIP_IM_FUSE, a developer switch (tests/test_gpu_implicit*.py): 0 = two launches as before;
// 2 = every instance is marked pending, 3 = every second one -- the hand-over that no seeded instance takes by itself.
template <class M>
static int fuse_mode(const DevOptions &o, const asif_hip_solver &S, bool assemble_only, bool rb)
{
	if (!(im_fuse_qp<M>::value && !rb && !assemble_only && S.polish == 2 && S.lanes_per_qp == 0)) return 0;
	// an input pinned by its bounds (lb == ub; the relaxation variables never are: their upper bound is "none") is
	// eliminated by the solver's entry before the method runs (gi_small.hpp) -- stage 2's business
	if (o.lb[0] == o.ub[0] || o.relaxLb >= o.inf || o.relaxReachLb >= o.inf) return 0;
	const char *v = getenv("ASIF_HIP_IM_FUSE");
	if (v && v[0] == '0') return 0;
	if (v && (v[0] == '2' || v[0] == '3')) return v[0] - '0';
	return 1;
}

int launch_implicit_ip(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream, bool rb)
{
	using M = InvertedPendulum;
	if (a.B <= 0) return 0;
	if (o.integrator == 1 && rb) return ASIF_HIP_EUNSUPPORTED; // held input: time-dependent rhs, Euler only
	FilterArgs ac = a;
	ac.compactRows = assemble_only ? 0 : 1; // the filter's own rows: three doubles each (launchers.hpp)
	ac.fuseQp = fuse_mode<M>(o, S, assemble_only, rb);
	int e = launch_rows<M>(o, ac, stream, rb);
	if (e || assemble_only) return e;
	const ImplicitPolicy<M> p = {a.B, o, ac};
	switch (S.lanes_per_qp) {
	case 0:
	case 8: return launch_policy<3, 41, 8>(S, p, stream);
	case 4: return launch_policy<3, 41, 4>(S, p, stream);
	case 16: return launch_policy<3, 41, 16>(S, p, stream);
	default: return ASIF_HIP_EINVAL;
	}
}

// examples/DoubleIntegrator_implicit.cpp: npBTSS = 4 -> nc = 4*4 + 1 = 17 rows
int launch_implicit_di(const DevOptions &o, const asif_hip_solver &S, const FilterArgs &a, bool assemble_only,
                       hipStream_t stream, bool rb)
{
	using M = DoubleIntegratorImplicit;
	static_assert(M::NPBTSS * M::NPSS + M::NPBS == 17, "QP shape 3 x 17");
	if (a.B <= 0) return 0;
	if (o.integrator == 1 && rb) return ASIF_HIP_EUNSUPPORTED;
	FilterArgs ac = a;
	ac.compactRows = assemble_only ? 0 : 1; // the filter's own rows: three doubles each (launchers.hpp)
	ac.fuseQp = fuse_mode<M>(o, S, assemble_only, rb);
	int e = launch_rows<M>(o, ac, stream, rb);
	if (e || assemble_only) return e;
	const ImplicitPolicy<M> p = {a.B, o, ac};
	switch (S.lanes_per_qp) {
	case 0:
	case 4: return launch_policy<3, 17, 4>(S, p, stream);
	case 2: return launch_policy<3, 17, 2>(S, p, stream);
	case 8: return launch_policy<3, 17, 8>(S, p, stream);
	default: return ASIF_HIP_EINVAL;
	}
}

} // namespace asif
