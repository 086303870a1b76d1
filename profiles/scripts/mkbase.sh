# build HEAD's library as build_ab/libbase.so (the working tree's build stays dumphfdl_amd/libhfdl_gpu.so)
set -e
cd /root/repo
rm -rf /tmp/basecsrc && mkdir -p /tmp/basecsrc/dumphfdl_amd/csrc /tmp/basecsrc/include build_ab
for f in $(git ls-files dumphfdl_amd/csrc include); do git show HEAD:$f > /tmp/basecsrc/$f; done
HFDL_OUT=/root/repo/build_ab/libbase.so HFDL_BUILD_DIR=/tmp/basecsrc/build bash /tmp/basecsrc/dumphfdl_amd/csrc/build.sh | tail -1
