"""
Error types of the Pyrado interface (Pyrado/pyrado/utils/exceptions.py:57-195): TypeErr, ValueErr, ShapeErr, PathErr.
They subclass the matching built-in exceptions and assemble their message from keyword arguments.
"""


def _name(obj):
    return type(obj).__name__


class BaseErr:
    @staticmethod
    def retrieve_var_name(var, stack_level: int = 2):
        return 'the input'


class TypeErr(TypeError):
    def __init__(self, *, given=None, given_name: str = None, expected_type=None, msg: str = None):
        if msg is None:
            exp = expected_type if isinstance(expected_type, (list, tuple)) else [expected_type]
            names = ' or '.join(getattr(t, '__name__', str(t)) for t in exp)
            msg = f'Expected the type of {given_name or "the input"} to be {names} but received {_name(given)}!'
        super().__init__(msg)


class ValueErr(ValueError):
    def __init__(self, *, given=None, given_name: str = None, eq_constraint=None, l_constraint=None, le_constraint=None,
                 g_constraint=None, ge_constraint=None, msg: str = None):
        if msg is None:
            parts = []
            for label, c in (('equal to', eq_constraint), ('less than', l_constraint),
                             ('less than or equal to', le_constraint), ('greater than', g_constraint),
                             ('greater than or equal to', ge_constraint)):
                if c is not None:
                    parts.append(f'{label} {c}')
            msg = f'The value of {given_name or "the input"} must be ' + ' and '.join(parts) + f', but it is {given}!'
        super().__init__(msg)


class ShapeErr(RuntimeError):
    def __init__(self, *, given=None, given_name: str = None, expected_match=None, msg: str = None):
        if msg is None:
            gs = getattr(given, 'shape', None)
            es = getattr(expected_match, 'shape', None)
            msg = f'The shape of {given_name or "the input"} {gs} does not match the expected shape {es}!'
        super().__init__(msg)


class PathErr(FileNotFoundError):
    def __init__(self, *, given=None, msg: str = None):
        super().__init__(msg if msg is not None else f'{given} is not a valid path!')
