class Variable:
    pass


class Constant(Variable):
    pass
