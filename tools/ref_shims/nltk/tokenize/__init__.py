def word_tokenize(text):
    raise RuntimeError('nltk is not available in this container; use --tokenizer MIND')
