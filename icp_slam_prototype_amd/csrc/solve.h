// solve.h -- host-side 3x3 solves of the ICP step (float64 arithmetic).
#pragma once
#include <stdint.h>

namespace icpk {

struct Mat3 {
  double m[3][3];
};

}  // namespace icpk
