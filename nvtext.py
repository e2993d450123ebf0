"""`import nvtext` -- the reference's module name (python/nvtext.py) for this back-end."""
from custrings_amd.nvtext import *  # noqa: F401,F403
from custrings_amd.nvtext import tokenize, ngrams, unique_tokens, token_count, tokens_counts, replace_tokens, normalize_spaces  # noqa: F401
