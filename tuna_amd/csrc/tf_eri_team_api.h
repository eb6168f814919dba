// tf_eri_team_api.h -- what tf_device.hip sees of the team ERI kernels (tf_eri_team.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "tf_dbasis.hip.h"

namespace tfk {

#define TF_TEAM_LMAX 6           // pair sums La + Lb, Lc + Ld the team kernels are instantiated for (up to two f shells)

struct TClass;                    // tf_eri_team.hip.h
struct BraRec;
struct KetRec;

struct TeamLaunch {
    int LAB, LCD, team;           // pair sums; lanes per shell quartet (eri_team_size)
    dim3 grid;                    // (ket groups of 256 / team pairs, bra pairs)
    size_t lds_bytes;
    hipStream_t stream;
    const DBasis *B;
    const TClass *tc;
    const BraRec *bras;           // one record per bra pair of the launch (grid.y)
    const KetRec *kets;           // the ket class list
    const int *kcnt;              // [shell A]: kets of the class list whose first shell is <= A
    double *T2;
};

struct TeamTask;                  // tf_eri_teamc.hip.h
struct TeamcLaunch {              // eri_teamc_kernel: one launch per (LAB, LCD, team) over a task list that may mix classes
    int LAB, LCD, team;
    unsigned n_tasks;
    size_t lds_bytes;             // the largest carve-out among the classes of the tasks
    hipStream_t stream;
    const DBasis *B;
    const TClass *tcs;            // device: class records
    const TeamTask *tasks;        // device: this launch's tasks
    const int *klist;             // device: ket pair ids the tasks point into
    int pq_max;                   // quartets with more primitive quartets are skipped (eri_cfact_kernel computes them)
    double *T2;
};
hipError_t eri_teamc_launch(const TeamcLaunch &a);

bool eri_team_available(int LAB, int LCD, int team);   // is this combination instantiated
hipError_t eri_team_launch(const TeamLaunch &a);

}  // namespace tfk
