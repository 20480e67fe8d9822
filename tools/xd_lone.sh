#!/bin/bash
# GPU box: how fast is the K loop of a LONE wave per SIMD?  Half-workgroup experiment (4 waves, no halo: timing only) with one
# workgroup per CU (LDS ballast): operand rings 4 / 8 / 11 deep, refills clustered after a diagonal's MFMAs or interleaved with them.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export XP=${XP:-110}
for r in 4 8 11; do echo "== lone wave, rings $r"; bash $R/tools/xd_stamp.sh -DXD_EXP_HALFWG -DXD_EXP_LONE -DXD_RA=$r -DXD_RB=$r 2>&1 | grep plane; done
for r in 5 8; do echo "== lone wave, rings $r, interleaved refills"; bash $R/tools/xd_stamp.sh -DXD_EXP_HALFWG -DXD_EXP_LONE -DXD_INTERLEAVE -DXD_RA=$r -DXD_RB=$r 2>&1 | grep plane; done
for r in 8; do echo "== two per CU, rings $r"; bash $R/tools/xd_stamp.sh -DXD_EXP_HALFWG -DXD_RA=$r -DXD_RB=$r 2>&1 | grep plane; done
