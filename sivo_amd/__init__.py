"""sivo_amd — MI355X-native implementation of navganti/SIVO's per-frame perception hot path
(Bayesian SegNet x T MC-dropout + entropy, ORB extraction, Hamming matching, BA edge
linearisation) behind the C ABI of include/sivo_hip.h.  See DESIGN.md."""
__version__ = "0.1.0"
