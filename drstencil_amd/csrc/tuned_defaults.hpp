// tuned_defaults.hpp -- GENERATED from drstencil_amd/tuned_defaults.tsv by drstencil_amd/tuned_defaults.py (tuning.py --write-defaults): do not edit.
// The tuner's winners per problem class; generator.hpp applies a row when a command line gives no geometry / emission option.
#pragma once
namespace drs {
struct TunedDefault { const char *mode; unsigned shape; int points, order, step; const char *dtype; int temporal, N; const char *options; };
static const TunedDefault kTunedDefaults[] = {
    {"2d", 0x51042f74u, 5, 1, 2, "fp64", 0, 8192, "--bx 64 --by 4 --block-merge-x 2 --block-merge-y 4 --xcd-remap 0"},
    {"2d", 0xe818b6ceu, 5, 1, 1, "fp32", 0, 8192, "--bx 128 --by 2 --block-merge-x 4 --block-merge-y 2 --xcd-remap 0"},
    {"2d", 0xe818b6ceu, 5, 1, 1, "fp64", 0, 8192, "--bx 128 --by 4 --block-merge-x 2 --block-merge-y 2 --xcd-remap 0"},
    {"2d", 0xe818b6ceu, 5, 1, 2, "fp64", 0, 8192, "--bx 128 --by 4 --block-merge-x 2 --block-merge-y 2 --xcd-remap 0"},
    {"2d", 0x610f3e78u, 9, 1, 2, "fp64", 0, 8192, "--bx 64 --by 4 --block-merge-x 2 --block-merge-y 6 --xcd-remap 0"},
    {"2d", 0x694374dcu, 9, 2, 2, "fp64", 0, 8192, "--bx 64 --by 4 --block-merge-x 2 --block-merge-y 4 --xcd-remap 0 --order rows"},
    {"2d", 0x7546f2ecu, 9, 2, 2, "fp64", 0, 8192, "--bx 64 --by 4 --block-merge-x 2 --block-merge-y 4 --xcd-remap 0 --order rows"},
    {"2d", 0x4e1e9130u, 25, 2, 1, "fp64", 0, 16384, "--bx 64 --by 4 --block-merge-x 2 --block-merge-y 8 --xcd-remap 0 --order rows"},
    {"2ds", 0x51042f74u, 5, 1, 2, "fp64", 0, 8192, "--bx 256 --by 1 --sn 16 --stream-unroll 4 --block-merge-x 2 --cyclic-merge-y 1 --merge-forward 5 --prefetch --prefetch-depth 1 --xrim dpp --xcd-remap 0 --schedule scatter"},
    {"2ds", 0xe818b6ceu, 5, 1, 2, "fp64", 0, 8192, "--bx 256 --by 1 --sn 16 --stream-unroll 4 --block-merge-x 2 --cyclic-merge-y 1 --merge-forward 5 --prefetch --prefetch-depth 1 --xrim dpp --xcd-remap 0 --schedule scatter"},
    {"2ds", 0x610f3e78u, 9, 1, 2, "fp64", 0, 8192, "--bx 256 --by 1 --sn 16 --stream-unroll 4 --block-merge-x 2 --cyclic-merge-y 1 --merge-forward 5 --prefetch --prefetch-depth 1 --xrim dpp --xcd-remap 0 --schedule scatter"},
    {"2ds", 0x694374dcu, 9, 2, 2, "fp64", 0, 8192, "--bx 128 --by 1 --sn 128 --stream-unroll 4 --block-merge-x 2 --cyclic-merge-y 1 --merge-forward 5 --xrim dpp --xcd-remap 0 --schedule scatter --pin 1"},
    {"2ds", 0x694374dcu, 9, 2, 2, "fp64", 1, 8192, "--bx 64 --by 1 --sn 32 --stream-unroll 4 --block-merge-x 2 --cyclic-merge-y 1 --merge-forward 5 --prefetch --prefetch-depth 1 --xrim dpp --xcd-remap 0 --skew 1 --exact-y 1 --schedule scatter --order rows --pack 0"},
    {"2ds", 0x7546f2ecu, 9, 2, 2, "fp64", 0, 8192, "--bx 128 --by 1 --sn 128 --stream-unroll 4 --block-merge-x 2 --cyclic-merge-y 1 --merge-forward 5 --xrim dpp --xcd-remap 0 --schedule scatter --pin 1"},
    {"2ds", 0x7546f2ecu, 9, 2, 2, "fp64", 1, 8192, "--bx 64 --by 1 --sn 32 --stream-unroll 4 --block-merge-x 2 --cyclic-merge-y 1 --merge-forward 5 --prefetch --prefetch-depth 1 --xrim dpp --xcd-remap 0 --skew 1 --exact-y 1 --schedule scatter --order rows --pack 0"},
    {"2ds", 0x4e1e9130u, 25, 2, 2, "fp64", 1, 8192, "--prefetch --prefetch-depth 1 --bx 128 --by 1 --block-merge-x 2 --cyclic-merge-y 1 --sn 32 --xcd-remap 0"},
    {"3d", 0x6ff9ee97u, 7, 1, 1, "fp32", 0, 512, "--prefetch --bx 32 --by 16 --block-merge-x 4 --block-merge-y 2 --sn 4 --xcd-remap 2"},
    {"3d", 0x6ff9ee97u, 7, 1, 1, "fp32", 0, 1024, "--prefetch --bx 256 --by 2 --block-merge-x 4 --block-merge-y 4 --sn 4 --xcd-remap 2"},
    {"3d", 0x6ff9ee97u, 7, 1, 2, "fp32", 0, 512, "--prefetch --prefetch-depth 3 --bx 32 --by 16 --block-merge-x 4 --block-merge-y 2 --sn 64 --xcd-remap 2 --cc-opt -fno-slp-vectorize"},
    {"3d", 0x6ff9ee97u, 7, 1, 2, "fp32", 0, 1024, "--bx 64 --by 16 --block-merge-x 4 --block-merge-y 2 --sn 16 --xcd-remap 2 --pin 1 --cc-opt -fno-slp-vectorize"},
    {"3d", 0x6ff9ee97u, 7, 1, 2, "fp32", 1, 512, "--prefetch --bx 66 --by 15 --block-merge-x 4 --block-merge-y 2 --sn 16 --xcd-remap 0"},
    {"3d", 0x6ff9ee97u, 7, 1, 2, "fp32", 1, 1024, "--prefetch --bx 66 --by 15 --block-merge-x 4 --block-merge-y 2 --sn 32 --xcd-remap 4"},
    {"3d", 0x6ff9ee97u, 7, 1, 2, "fp64", 0, 512, "--bx 128 --by 4 --block-merge-x 2 --block-merge-y 2 --sn 32 --xcd-remap 2 --cc-opt -fno-slp-vectorize"},
    {"3d", 0x6ff9ee97u, 7, 1, 2, "fp64", 0, 1024, "--bx 64 --by 16 --block-merge-x 2 --block-merge-y 2 --sn 16 --xcd-remap 2 --pin 1 --cc-opt -fno-slp-vectorize"},
    {"3d", 0x6ff9ee97u, 7, 1, 3, "fp64", 1, 512, "--skew 2 --pin 1 --exact-y 1 --prefetch --bx 66 --by 15 --block-merge-x 2 --block-merge-y 2 --sn 64 --xcd-remap 4"},
    {"3d", 0x6ff9ee97u, 7, 1, 3, "fp64", 1, 1024, "--skew 1 --pin 1 --exact-y 1 --prefetch --bx 66 --by 15 --block-merge-x 2 --block-merge-y 2 --sn 256 --xcd-remap 4"},
    {"3d", 0x6ff9ee97u, 7, 1, 4, "fp64", 1, 512, "--skew 1 --pin 1 --exact-y 1 --prefetch --prefetch-depth 2 --bx 68 --by 11 --block-merge-x 2 --block-merge-y 2 --sn 64 --xcd-remap 4"},
    {"3d", 0x6ff9ee97u, 7, 1, 4, "fp64", 1, 1024, "--skew 1 --pin 1 --exact-y 1 --prefetch --prefetch-depth 2 --bx 68 --by 11 --block-merge-x 2 --block-merge-y 2 --sn 256 --xcd-remap 4"},
    {"3d", 0x62e6907cu, 9, 1, 2, "fp64", 0, 512, "--schedule scatter --prefetch --prefetch-depth 1 --bx 32 --by 16 --block-merge-x 2 --block-merge-y 2 --sn 32 --xcd-remap 2 --order rows"},
};
static const int kTunedDefaultsCount = (int)(sizeof(kTunedDefaults) / sizeof(kTunedDefaults[0]));
}  // namespace drs
