"""MI355X-native optical-flow -> ego-velocity hot path (HIP kernels behind a C ABI, numpy + ctypes host side).

Modules
  ofk              ctypes binding of libofk.so (include/ofk.h); fails loudly without the library / a gfx950 GPU
  cv2_hip          the cv2 calls of the reference's hot path (cvtColor, goodFeaturesToTrack, calcOpticalFlowPyrLK,
                   KalmanFilter) with OpenCV's signatures and array conventions, running on the HIP kernels
  of_library       drop-in for the reference's of_library.py (same names and positional signatures)
  velocity_node    generate_test_data / solve_lgs / optical_fusion of the reference's velocity_measurment_node
  simulation       generate_test_data / solve_lgs / feasibility / of_simulation of numerical_simulation/simulation.py
  pipeline         batched resident frame-pair pipeline (the benchmarked path)
  synth            synthetic frame-pair renderer (measurement harness)
"""
__version__ = "0.1.0"
