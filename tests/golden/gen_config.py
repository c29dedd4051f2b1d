"""Records the scalar defaults of the reference Config (configure.py:7-70) as data: tests/golden/config_defaults.json."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

if __name__ == "__main__":
    c = ref_harness.load_reference().Config()
    ref = {k: v for k, v in vars(c).items() if isinstance(v, (int, float, bool, str))}
    ref["temperature_at"] = {str(s): c.epsilon_by_frame(s) for s in (0, 1, 5, 10, 30, 119)}
    with open(os.path.join(HERE, "config_defaults.json"), "w") as f:
        json.dump(ref, f, indent=1, sort_keys=True)
    print(len(ref), "entries")
