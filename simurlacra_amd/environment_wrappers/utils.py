"""
Helpers to walk a chain of environment wrappers (Pyrado/pyrado/environment_wrappers/utils.py:11-163).
"""
from ..environments.base import Env
from .base import EnvWrapper


def all_envs(env: Env):
    yield env
    while isinstance(env, EnvWrapper):
        env = env.wrapped_env
        yield env


def inner_env(env: Env) -> Env:
    for e in all_envs(env):
        pass
    return e


def typed_env(env: Env, tp):
    """ Outermost environment of the chain that is an instance of `tp`, or None. """
    for e in all_envs(env):
        if isinstance(e, tp):
            return e
    return None


def attr_env(env: Env, attr_name: str):
    for e in all_envs(env):
        if hasattr(e, attr_name):
            return e
    return None


def attr_env_get(env: Env, attr_name: str):
    for e in all_envs(env):
        if hasattr(e, attr_name):
            return getattr(e, attr_name)
    return None


def remove_env(env: Env, tp):
    """ Remove the outermost wrapper of type `tp` from the chain and return the new outermost environment. """
    if isinstance(env, tp):
        return env.wrapped_env
    prev = env
    while isinstance(prev, EnvWrapper):
        if isinstance(prev.wrapped_env, tp):
            prev._wrapped_env = prev.wrapped_env.wrapped_env
            return env
        prev = prev.wrapped_env
    raise RuntimeError(f'no wrapper of type {tp} in the chain')
