def grad_not_implemented(op, idx, var, comment=""):
    return None
