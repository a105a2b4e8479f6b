"""Dev tool: timeline of the f32 fused joint kernels from a -DJH_TRACE build (JH_TRACE_FILE32)."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.int64)
p1 = a[:512].reshape(8, 64)
p2 = a[512:768].reshape(4, 64)
if p1.any():
    print("phase 1 (8 waves): C^T staging", (p1[:, 1] - p1[:, 0]).tolist())
    for w in (0, 7):
        print("  wave", w, "per row iteration (W2/MFMA loop, lsm epilogue, park + barrier):",
              [(int(p1[w, 2 + 4 * i] - p1[w, 1 + 4 * i]), int(p1[w, 3 + 4 * i] - p1[w, 2 + 4 * i]), int(p1[w, 4 + 4 * i] - p1[w, 3 + 4 * i])) for i in range(1, 5)])
if p2.any():
    for w in (0, 3):
        print("phase 2 wave", w, "per row iteration (dl row -> LDS + next fetch issue, MFMAs + epilogue):",
              [(int(p2[w, 3 * i + 1] - p2[w, 3 * i]), int(p2[w, 3 * i + 3] - p2[w, 3 * i + 1])) for i in range(2, 12)])
    print("  phase 2 total per wave", (p2[:, 60] - p2[:, 0]).tolist())
