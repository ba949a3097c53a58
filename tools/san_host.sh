#!/bin/bash
# tools/san_host.sh -- AddressSanitizer + UBSan over the host-only code (grid builder, camera and
# scene helpers, the headless argument parser).  No GPU needed; GPU code is not instrumented.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/san
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off -Iinclude -Iraytracer_amd/csrc \
    tools/san_host.cpp raytracer_amd/csrc/luma_grid.cpp raytracer_amd/csrc/luma_host.cpp -o build/san/san_test
./build/san/san_test
(cd raytracer_amd/host && g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -I../../include -I. \
    -o ../../build/san/headless_san main.cpp renderer.cpp camera.cpp arguments.cpp image.cpp log.cpp \
    -L.. -lluma_hip -lrt -Wl,-rpath,"$PWD/.." -Wl,-rpath,/opt/rocm/lib)
set +e
./build/san/headless_san --context=directx --width=abc --bogus=1 stray --mode=fancy --paths=3 --height --devices=0,,1 \
    --devices=1,2,3 --pos=1,2 --pos=1,2,3 --pitch=x --yaw=0.5 --seed=0xZZ --seed=0x10 --gpus=9 --gpus 4 \
    --frames=99999999999 --output= --shm= > build/san/parser.log 2>&1
rc=$?; set -e
test $rc -eq 2 || { cat build/san/parser.log; exit 1; }
echo "headless parser under sanitizers ok"
