#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.
#
# Builds oracle/_ref/libref.so from the reference's own hot-path sources, read from where
# they lie under $REF (default /root/reference), plus oracle/ref_harness.cpp.
# Nothing of the reference is copied into this repository: the portability-patched copies
# live in a temporary directory that is removed at exit, and the only output is the shared
# library in oracle/_ref/ (git-ignored).
#
# The patch is mechanical and only spells MSVC-isms the way g++ accepts them:
#   x.m128_f32            -> ((float*)&x)          (src/BVH.cpp:27-31, src/Triangle.cpp:51-82)
#   std::fabsf/std::powf  -> fabsf/powf            (src/Triangle.cpp:51-70, src/Texture.cpp:49-51)
#   std::exception("..")  -> std::runtime_error    (src/RayTracer.cpp:47,290-513)
# Standard headers the reference forgets to include are force-included (-include).
# No third-party library is stubbed: render()/loadModel()/parseConfigFile(), the only users
# of TBB and assimp, are dropped by --gc-sections because the harness never references
# them, and -z defs proves nothing undefined is left.
#
# Floating-point mode of the oracle: -O2 -mavx2 -mfma -ffp-contract=off, no fast-math.
set -euo pipefail
REF="${REF:-/root/reference}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
	echo "build_ref.sh: $REF/src not present; keeping any prebuilt $OUT/libref.so" >&2
	exit 0
fi
TMP="$(mktemp -d /tmp/rtref.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"
CXX="${CXX:-g++}"
CF="-std=c++17 -O2 -mavx2 -mfma -ffp-contract=off -fPIC -fvisibility=hidden -ffunction-sections -fdata-sections"
INC="-I$REF/include -include memory -include vector -include array -include cmath -include cfloat -include cstdint -include stdexcept -include immintrin.h -include Eigen/Geometry"
OBJS=""
for f in Camera BVH Triangle Texture Skybox ImageIO RayTracer; do
	sed -E \
		-e 's/([A-Za-z0-9_]+(\[[^]]+\])?)\.m128_f32/((float*)\&\1)/g' \
		-e 's/std::exception\(/std::runtime_error(/g' \
		-e 's/std::fabsf/fabsf/g' \
		-e 's/std::powf/powf/g' \
		"$REF/src/$f.cpp" > "$TMP/$f.cpp"
	$CXX $CF $INC -w -c "$TMP/$f.cpp" -o "$TMP/$f.o" &
	OBJS="$OBJS $TMP/$f.o"
done
$CXX $CF $INC -w -c "$HERE/ref_harness.cpp" -o "$TMP/ref_harness.o" &
wait
$CXX -shared -o "$OUT/libref.so" "$TMP/ref_harness.o" $OBJS -Wl,--gc-sections -Wl,-z,defs -lpthread
echo "built $OUT/libref.so"
