// Python binding for the reference's own CPU operators (ROIAlign forward, NMS), used ONLY to pin the CPU oracle
// (oracle/roi_oracle.py, oracle/nms_oracle.py) and to record tests/golden/roi_golden.npz.  Test infrastructure: nothing in
// the product path loads this module.  The two functions are declared by the reference's own header (csrc/cpu/vision.h,
// on the include path of oracle/build_ref.py); this file only exposes them to Python the way the reference's
// csrc/vision.cpp:7-15 does for `disprcnn._C`.
#include "cpu/vision.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("roi_align_forward", &ROIAlign_forward_cpu, "reference csrc/cpu/ROIAlign_cpu.cpp:220-257");
    m.def("nms", &nms_cpu, "reference csrc/cpu/nms_cpu.cpp:67-75");
}
