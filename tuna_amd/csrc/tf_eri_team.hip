// tf_eri_team.hip -- translation unit of the team ERI kernels (tf_eri_team.hip.h): the instantiations for the pair sums
// LAB, LCD = 0 .. TF_TEAM_LMAX and the three team sizes, and the run-time dispatch onto them.
#include <hip/hip_runtime.h>
#include "tf_eri_team.hip.h"
#include "tf_eri_team_api.h"

namespace tfk {

bool eri_team_available(int LAB, int LCD, int team)
{
    const int mn = (LAB + 1) * (LCD + 1);
    const int mx = (LAB / 2 + 1) * ((LAB + 1) / 2 + 1) * (LCD / 2 + 1) * ((LCD + 1) / 2 + 1);
    if (LAB > TF_TEAM_LMAX || LCD > TF_TEAM_LMAX) return false;
    return team == 16 ? mn <= 16 : (team == 64 ? mn <= 256 : (team == 256 && mx > 32));
}

template <int LAB, int LCD, int TEAM>
constexpr bool team_combo_possible()
{
    constexpr int mn = (LAB + 1) * (LCD + 1);                                       // one shell of each pair is an s shell
    constexpr int mx = (LAB / 2 + 1) * ((LAB + 1) / 2 + 1) * (LCD / 2 + 1) * ((LCD + 1) / 2 + 1);
    return TEAM == 16 ? mn <= 16 : (TEAM == 64 ? mn <= 256 : mx > 32);
}

template <int LAB, int LCD, int TEAM>
static hipError_t launch_one(const TeamLaunch &a)
{
    if constexpr (team_combo_possible<LAB, LCD, TEAM>()) {
        static size_t lds_set = 64 * 1024;
        if (a.lds_bytes > lds_set) {
            hipError_t e = hipFuncSetAttribute((const void *)eri_team_kernel<LAB, LCD, TEAM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            if (e != hipSuccess) return e;
            lds_set = 160 * 1024;
        }
        hipLaunchKernelGGL((eri_team_kernel<LAB, LCD, TEAM>), a.grid, dim3(256), a.lds_bytes, a.stream, *a.B, *a.tc, a.bras, a.kets, a.kcnt, a.T2);
        return hipGetLastError();
    } else
        return hipErrorInvalidValue;
}

template <int LAB, int LCD>
static hipError_t launch_team(const TeamLaunch &a)
{
    switch (a.team) {
    case 16: return launch_one<LAB, LCD, 16>(a);
    case 64: return launch_one<LAB, LCD, 64>(a);
    case 256: return launch_one<LAB, LCD, 256>(a);
    }
    return hipErrorInvalidValue;
}

template <int LAB>
static hipError_t launch_lcd(const TeamLaunch &a)
{
    switch (a.LCD) {
    case 0: return launch_team<LAB, 0>(a);
    case 1: return launch_team<LAB, 1>(a);
    case 2: return launch_team<LAB, 2>(a);
    case 3: return launch_team<LAB, 3>(a);
    case 4: return launch_team<LAB, 4>(a);
    case 5: return launch_team<LAB, 5>(a);
    case 6: return launch_team<LAB, 6>(a);
    }
    return hipErrorInvalidValue;
}

hipError_t eri_team_launch(const TeamLaunch &a)
{
    switch (a.LAB) {
    case 0: return launch_lcd<0>(a);
    case 1: return launch_lcd<1>(a);
    case 2: return launch_lcd<2>(a);
    case 3: return launch_lcd<3>(a);
    case 4: return launch_lcd<4>(a);
    case 5: return launch_lcd<5>(a);
    case 6: return launch_lcd<6>(a);
    }
    return hipErrorInvalidValue;
}

}  // namespace tfk
