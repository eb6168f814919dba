// tf_eri_teamc.hip -- translation unit of the contracted / task-list team ERI kernels (tf_eri_teamc.hip.h).
#include <hip/hip_runtime.h>
#include "tf_eri_teamc.hip.h"
#include "tf_eri_team_api.h"

namespace tfk {

template <int LAB, int LCD, int TEAM>
static hipError_t launchc_one(const TeamcLaunch &a)
{
    constexpr int mn = (LAB + 1) * (LCD + 1);
    constexpr int mx = (LAB / 2 + 1) * ((LAB + 1) / 2 + 1) * (LCD / 2 + 1) * ((LCD + 1) / 2 + 1);
    if constexpr (TEAM == 16 ? mn <= 16 : (TEAM == 64 ? mn <= 256 : mx > 32)) {      // the combinations eri_team_available() reports
        static size_t lds_set = 64 * 1024;
        if (a.lds_bytes > lds_set) {
            hipError_t e = hipFuncSetAttribute((const void *)eri_teamc_kernel<LAB, LCD, TEAM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            if (e != hipSuccess) return e;
            lds_set = 160 * 1024;
        }
        hipLaunchKernelGGL((eri_teamc_kernel<LAB, LCD, TEAM>), dim3(a.n_tasks), dim3(256), a.lds_bytes, a.stream, *a.B, a.tcs, a.tasks, a.klist, a.pq_max, a.T2);
        return hipGetLastError();
    } else
        return hipErrorInvalidValue;
}

template <int LAB, int LCD>
static hipError_t launchc_team(const TeamcLaunch &a)
{
    switch (a.team) {
    case 16: return launchc_one<LAB, LCD, 16>(a);
    case 64: return launchc_one<LAB, LCD, 64>(a);
    case 256: return launchc_one<LAB, LCD, 256>(a);
    }
    return hipErrorInvalidValue;
}

template <int LAB>
static hipError_t launchc_lcd(const TeamcLaunch &a)
{
    switch (a.LCD) {
    case 0: return launchc_team<LAB, 0>(a);
    case 1: return launchc_team<LAB, 1>(a);
    case 2: return launchc_team<LAB, 2>(a);
    case 3: return launchc_team<LAB, 3>(a);
    case 4: return launchc_team<LAB, 4>(a);
    case 5: return launchc_team<LAB, 5>(a);
    case 6: return launchc_team<LAB, 6>(a);
    }
    return hipErrorInvalidValue;
}

hipError_t eri_teamc_launch(const TeamcLaunch &a)
{
    switch (a.LAB) {
    case 0: return launchc_lcd<0>(a);
    case 1: return launchc_lcd<1>(a);
    case 2: return launchc_lcd<2>(a);
    case 3: return launchc_lcd<3>(a);
    case 4: return launchc_lcd<4>(a);
    case 5: return launchc_lcd<5>(a);
    case 6: return launchc_lcd<6>(a);
    }
    return hipErrorInvalidValue;
}

}  // namespace tfk
