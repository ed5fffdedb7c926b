// solve.cpp -- the implementations live in solve_impl.h (shared host/device); this
// translation unit only anchors them for the host side of libicpk.so.
#include "solve_impl.h"
