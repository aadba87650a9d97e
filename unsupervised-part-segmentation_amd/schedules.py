"""Schedule variables of the yaml surface (edflow ``make_var`` / ``make_linear_var``; the formulas are
mirrored in-tree at cub/code/nn.py:1064-1083)."""


def make_linear_var(step, start, end, start_value, end_value, clip_min=0.0, clip_max=1.0):
    v = (end_value - start_value) / (end - start) * (float(step) - start) + start_value
    return float(min(max(v, clip_min), clip_max))


def make_staircase_var(step, start, start_value, step_size, stair_factor, clip_min=0.0, clip_max=1.0):
    v = stair_factor ** ((float(step) - start) // step_size) * start_value
    return float(min(max(v, clip_min), clip_max))


def make_var(step, spec):
    if spec["var_type"] == "linear":
        return make_linear_var(step, **spec["options"])
    if spec["var_type"] == "staircase":
        return make_staircase_var(step, **spec["options"])
    raise ValueError("unknown var_type {}".format(spec["var_type"]))
