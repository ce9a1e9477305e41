// lds_table.cpp -- lookup of the per-instance launchers defined in lds_inst.hip objects.
#include "lds_launch.h"

namespace fhe_dev {

#define DECL(F, N) void lds_launch_##F##_##N(const LdsArgs &);
DECL(F32, 11) DECL(F32, 12) DECL(F32, 13) DECL(F32, 14) DECL(F32, 15)
DECL(F64, 11) DECL(F64, 12) DECL(F64, 13) DECL(F64, 14)
DECL(F64X, 11) DECL(F64X, 12) DECL(F64X, 13) DECL(F64X, 14)
DECL(F52, 11) DECL(F52, 12) DECL(F52, 13) DECL(F52, 14)
#undef DECL

lds_launch_fn lds_lookup(int width, int log_n) {
    if (width == 32) {
        switch (log_n) {
            case 11: return lds_launch_F32_11; case 12: return lds_launch_F32_12; case 13: return lds_launch_F32_13;
            case 14: return lds_launch_F32_14; case 15: return lds_launch_F32_15;
        }
    } else if (width == 52) {
        switch (log_n) {
            case 11: return lds_launch_F52_11; case 12: return lds_launch_F52_12; case 13: return lds_launch_F52_13;
            case 14: return lds_launch_F52_14;
        }
    } else if (width == 64) {
        switch (log_n) {
            case 11: return lds_launch_F64_11; case 12: return lds_launch_F64_12; case 13: return lds_launch_F64_13;
            case 14: return lds_launch_F64_14;
        }
    } else if (width == 65) {     // F64X: full-range 64-bit primes
        switch (log_n) {
            case 11: return lds_launch_F64X_11; case 12: return lds_launch_F64X_12; case 13: return lds_launch_F64X_13;
            case 14: return lds_launch_F64X_14;
        }
    }
    return nullptr;
}

}  // namespace fhe_dev
