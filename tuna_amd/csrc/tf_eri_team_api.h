// tf_eri_team_api.h -- what tf_device.hip sees of the team ERI kernels (tf_eri_team.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "tf_dbasis.hip.h"

namespace tfk {

#define TF_TEAM_LMAX 6           // pair sums La + Lb, Lc + Ld the team kernels are instantiated for (up to two f shells)

struct TClass;                    // tf_eri_team.hip.h

struct TeamLaunch {
    int LAB, LCD, team;           // pair sums; lanes per shell quartet (eri_team_size)
    dim3 grid;                    // (ket groups of 256 / team pairs, bra pairs)
    size_t lds_bytes;
    hipStream_t stream;
    const DBasis *B;
    const TClass *tc;
    const int *bra_pairs;
    const long long *bra_rowoff;
    const int *ket_pairs;
    double *T2;
};

int eri_team_size(int nT);
hipError_t eri_team_launch(const TeamLaunch &a);

}  // namespace tfk
