"""
simurlacra_amd — MI355X-native batched rollout engine behind the Pyrado API for the standalone-Python
simulations (SimPyEnv subclasses).  The numerics run in hand-written HIP kernels (libpysimhip.so); this
package is the host-side mirror of the Pyrado interface for that path.
"""
inf = float('inf')
