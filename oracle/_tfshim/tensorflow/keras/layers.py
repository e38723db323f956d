from ._impl import (Layer, InputSpec, Dense, Conv1D, Conv2D, SeparableConv1D, LayerNormalization, BatchNormalization, Dropout,   # noqa: F401
                    Add, Activation, LeakyReLU, ReLU, Softmax, Embedding, Reshape, AveragePooling1D, MaxPool1D, MaxPooling1D,
                    MultiHeadAttention, SimpleRNN, LSTM, GRU)
