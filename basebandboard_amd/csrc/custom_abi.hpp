// custom_abi.hpp -- the one constant a per-matrix kernel library (csrc/custom_fill_template.hip) shares with libbbb_hip.so.
#pragma once
// Version of what a custom library (csrc/custom_fill_template.hip, built per matrix) shares with this one: the bit-plane
// layout it is handed, TrialDev, the generator numbering.  Bumped whenever any of them changes;
// bbb_lutopt_attach_custom_library refuses a library that reports another value (or none).
#define BBB_CUSTOM_ABI 4

