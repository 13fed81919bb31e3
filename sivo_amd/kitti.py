"""KITTI sequence listing and trajectory writer — the file formats at the two ends of SIVO's frame loop
(reference src/sivo.cc:145-177 loadImages; src/orbslam/System.cc:322-329 SaveTrajectoryKITTI line format).
Mirrors sivo_amd/api/kitti_io.hpp."""
import os

import numpy as np


def load_images(path_to_sequence):
    """-> (left file names, right file names, timestamps): times.txt, image_2/%06d.png, image_3/%06d.png."""
    times = []
    with open(os.path.join(path_to_sequence, "times.txt")) as f:
        for line in f:
            line = line.strip()
            if line:
                times.append(float(line.split()[0]))
    left = [f"{path_to_sequence}/image_2/{i:06d}.png" for i in range(len(times))]
    right = [f"{path_to_sequence}/image_3/{i:06d}.png" for i in range(len(times))]
    return left, right, times


def save_trajectory_kitti(filename, Tcw):
    """Tcw: (n, 12) float32 (Rcw row-major, tcw).  Writes the 3x4 [Rwc | twc] per line, `fixed`, 9 decimals."""
    Tcw = np.asarray(Tcw, np.float32).reshape(-1, 12)
    with open(filename, "w") as f:
        for T in Tcw:
            Rcw = T[:9].reshape(3, 3); tcw = T[9:]
            Rwc = Rcw.T
            twc = -(Rwc[:, 0] * tcw[0] + Rwc[:, 1] * tcw[1] + Rwc[:, 2] * tcw[2]).astype(np.float32)
            row = []
            for i in range(3):
                row += [Rwc[i, 0], Rwc[i, 1], Rwc[i, 2], twc[i]]
            f.write(" ".join(f"{float(v):.9f}" for v in row) + "\n")
