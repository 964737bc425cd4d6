"""Import stand-in for torchtext (absent here): only GloVe's `.stoi` / `.vectors` attributes are read by MIND_corpus.py."""
