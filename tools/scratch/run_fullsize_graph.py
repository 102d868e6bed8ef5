import sys
sys.path.insert(0, '.')
import tests.test_fullsize_gpu as t
big = t.big.__wrapped__() if hasattr(t.big, "__wrapped__") else None
t.test_fullsize_step_deterministic_and_graph_equals_eager(big)
print("OK")
