// TEST-ONLY: host stand-in for dtqn_amd/csrc/dtqn_gfx950.hpp.  The emulation's <hip/hip_runtime.h> already defines the
// host equivalents of every primitive (lane swaps, row rotates, agent-scope loads / stores, hand-over descriptors, fences).
#pragma once
#include <hip/hip_runtime.h>
