from .context import Context, get_context, set_context, reset_context
