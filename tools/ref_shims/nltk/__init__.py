"""Import stand-in for nltk (absent here): the reference imports it at module level but the MIND tokenizer path never calls it."""
