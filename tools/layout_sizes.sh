for sz in "1920 1080" "3840 2160" "7680 4320"; do set -- $sz; echo "== $1 x $2"; python tools/k4_ab.py --mode decode --layouts 1,2 --width $1 --height $2 x 2>&1 | grep -v amdgpu.ids; done
