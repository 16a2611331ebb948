#!/usr/bin/env bash
# Compiles the REFERENCE's own InfiniTAM/ITMLib CPU engine (sources stay where they lie under
# /root/reference, nothing is copied) plus oracle/ref_driver.cpp into oracle/_ref/itm_ref.
# Outputs only go to oracle/_ref/ (git-ignored, but travels to the GPU box with the snapshot).
# Deterministic build: no OpenMP, -ffp-contract=off, so hash-collision resolution follows pixel
# scan order and fp32 rounding is the plain IEEE sequence the C restatement (tsdf_oracle.c) uses.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${GPS_REFERENCE_ROOT:-/root/reference}/InfiniTAM"
OUT="$HERE/_ref"
mkdir -p "$OUT/obj"
[ -d "$REF" ] || { echo "reference not present: $REF" >&2; exit 0; }
if [ -x "$OUT/itm_ref" ] && [ -x "$OUT/itm_ref_omp" ] && [ "$OUT/itm_ref_omp" -nt "$HERE/ref_driver.cpp" ] && [ "$OUT/itm_ref_omp" -nt "$HERE/ref_build.sh" ]; then exit 0; fi
CXX="${CXX:-g++}"
FLAGS="-O2 -std=c++17 -DCOMPILE_WITHOUT_CUDA -ffp-contract=off -w -I$REF"
SRCS=(
  $REF/ITMLib/CPUInstantiations.cpp
  $REF/ITMLib/Engines/LowLevel/*Factory.cpp $REF/ITMLib/Engines/LowLevel/CPU/*.cpp
  $REF/ITMLib/Engines/ViewBuilding/*Factory.cpp $REF/ITMLib/Engines/ViewBuilding/CPU/*.cpp
  $REF/ITMLib/Engines/Visualisation/Interface/*.cpp
  $REF/ITMLib/Objects/Camera/*.cpp $REF/ITMLib/Objects/RenderStates/*.cpp
  $REF/ITMLib/Trackers/CPU/*.cpp $REF/ITMLib/Trackers/Interface/*.cpp
  $REF/ITMLib/Utils/*.cpp
  $REF/ITMLib/Engines/MultiScene/*.cpp
  $REF/ORUtils/*.cpp
  $REF/FernRelocLib/FernConservatory.cpp $REF/FernRelocLib/PoseDatabase.cpp $REF/FernRelocLib/RelocDatabase.cpp
  $REF/MiniSlamGraphLib/*.cpp
)
pids=()
objs=()
i=0
for s in "${SRCS[@]}"; do
  o="$OUT/obj/$(echo "$s" | md5sum | cut -c1-8)_$(basename "${s%.cpp}").o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ]; then
    $CXX $FLAGS -c "$s" -o "$o" &
    pids+=($!)
    i=$((i+1))
    if [ $((i % 8)) -eq 0 ]; then wait; fi
  fi
done
wait
$CXX $FLAGS -c "$HERE/ref_driver.cpp" -o "$OUT/obj/ref_driver.o"
$CXX -o "$OUT/itm_ref" "$OUT/obj/ref_driver.o" "${objs[@]}" -lpthread
echo "built $OUT/itm_ref"

# Second binary, built the way upstream builds its CPU path (InfiniTAM/CMakeLists.txt:25, cmake/UseOpenMP.cmake:5-13:
# -O3 + OpenMP; no -march=native because the binary travels to another host): only used by bench.py's cpu_baseline
# (timing mode of ref_driver), never by parity tests.
OFLAGS="-O3 -std=c++17 -DCOMPILE_WITHOUT_CUDA -DWITH_OPENMP -fopenmp -w -I$REF"
mkdir -p "$OUT/obj_omp"
oobjs=()
i=0
for s in "${SRCS[@]}"; do
  o="$OUT/obj_omp/$(echo "$s" | md5sum | cut -c1-8)_$(basename "${s%.cpp}").o"
  oobjs+=("$o")
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ]; then
    $CXX $OFLAGS -c "$s" -o "$o" &
    i=$((i+1))
    if [ $((i % 8)) -eq 0 ]; then wait; fi
  fi
done
wait
$CXX $OFLAGS -c "$HERE/ref_driver.cpp" -o "$OUT/obj_omp/ref_driver.o"
$CXX -fopenmp -o "$OUT/itm_ref_omp" "$OUT/obj_omp/ref_driver.o" "${oobjs[@]}" -lpthread
echo "built $OUT/itm_ref_omp"
