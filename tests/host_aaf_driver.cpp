// host_aaf_driver.cpp -- test driver (not product): replays the instruction programs of
// tests/golden/affa_programs.json on the HOST affine arithmetic of asif_amd/host/include/asif_affine.h (the
// `AAF` / `interval` the C++ mirror classes hand to user callbacks) and prints, per register, centre, bounds and
// coefficients, so that tests/test_host_affine.py can compare them with what the reference's libaffa produced.
//   stdin:  nprog nreg, then nprog lines "op dst a b imm0 imm1" (imm as hex floats)
//   stdout: per register "center lo hi n c0 c1 ..." (hex floats)
#include <cstdio>
#include <vector>
#include "asif_affine.h"

int main()
{
	int nprog, nreg;
	if (std::scanf("%d %d", &nprog, &nreg) != 2) return 2;
	AAF::set_default(0);
	std::vector<AAF> R(nreg);
	for (int p = 0; p < nprog; p++) {
		int op, d, a, b;
		double i0, i1;
		if (std::scanf("%d %d %d %d %la %la", &op, &d, &a, &b, &i0, &i1) != 6) return 3;
		switch (op) { // the op codes of oracle/ref_affa_shim.cpp
		case 0: R[d] = AAF(i0); break;
		case 1: R[d] = AAF(interval(i0, i1)); break;
		case 2: R[d] = R[a] + R[b]; break;
		case 3: R[d] = R[a] - R[b]; break;
		case 4: R[d] = R[a] * R[b]; break;
		case 5: R[d] = R[a] / R[b]; break;
		case 6: R[d] = inv(R[a]); break;
		case 7: R[d] = -R[a]; break;
		case 8: R[d] = R[a] * i0; break;
		case 9: R[d] = sin(R[a]); break;
		case 10: R[d] = R[a]; break;
		default: return 4;
		}
	}
	for (int r = 0; r < nreg; r++) {
		const interval iv = R[r].convert();
		std::printf("%a %a %a %u", R[r].get_center(), iv.left(), iv.right(), R[r].get_length());
		for (unsigned k = 0; k < R[r].get_length(); k++) std::printf(" %a", R[r].get_coeff(k));
		std::printf("\n");
	}
	return 0;
}
