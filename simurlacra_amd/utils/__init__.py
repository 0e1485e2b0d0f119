def get_class_name(obj) -> str:
    return obj.__class__.__name__
