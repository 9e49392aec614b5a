#!/bin/bash
# is there any route to a real OpenCV on the GPU lease?  Output is committed as profiles/r04_opencv_probe.txt.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4probe
mkdir -p $O
{
  echo "== date: $(date -u)"; echo "== host: $(uname -a)"
  echo "== python -c 'import cv2'"; python -c "import cv2; print(cv2.__version__)" 2>&1 | tail -2
  echo "== python3 -m pip --version"; python3 -m pip --version 2>&1 | tail -1
  echo "== pip download opencv-python-headless (20 s limit)"
  (cd /tmp && timeout 20 python3 -m pip download --no-deps -d /tmp/cvdl opencv-python-headless 2>&1 | tail -4); echo "rc=$?"
  echo "== pip index / wheelhouse"; python3 -m pip config list 2>&1 | tail -5; ls /opt/wheelhouse /root/wheelhouse /wheelhouse 2>&1 | head -5
  echo "== find libopencv* / cv2* / opencv2 headers"
  find / -xdev \( -name 'libopencv*' -o -name 'cv2*' -o -name 'opencv2' -o -name 'opencv*.whl' -o -name 'OpenCVConfig*.cmake' \) 2>/dev/null | head -20
  echo "(end of find)"
  echo "== conda"; ls /opt/conda/bin/conda 2>&1; /opt/conda/bin/conda list 2>/dev/null | grep -i -E 'opencv|ceres|eigen' ; ls /opt/conda/pkgs 2>/dev/null | grep -i -E 'opencv|ceres|eigen'
  echo "== apt"; apt list --installed 2>/dev/null | grep -i -E 'opencv|ceres|eigen' | head; timeout 15 apt-get download libopencv-dev 2>&1 | tail -2
  echo "== network"; timeout 5 python3 -c "import socket; socket.create_connection(('pypi.org',443),3); print('pypi reachable')" 2>&1 | tail -1
  echo "== other python interpreters"; ls /usr/bin/python3* /opt/conda/bin/python* 2>/dev/null; for p in /opt/conda/bin/python; do [ -x $p ] && $p -c "import cv2; print('conda cv2', cv2.__version__)" 2>&1 | tail -1; done
  echo "== skimage / PIL / kornia (independent implementations of CLAHE etc.)"; python3 - <<'PY'
for m in ("skimage", "PIL", "kornia", "torchvision", "imageio", "scipy"):
    try:
        mod = __import__(m); print(m, getattr(mod, "__version__", "?"))
    except Exception as e:
        print(m, "absent:", type(e).__name__)
PY
} > $O/opencv_probe.txt 2>&1
cat $O/opencv_probe.txt
