"""LangChain integration (reference: python/llm/src/ipex_llm/langchain/).  The classes derive from LangChain's base
classes when ``langchain_core`` is importable and from minimal local stand-ins with the same call surface
(``invoke`` / ``__call__`` / ``embed_documents`` / ``embed_query``) otherwise."""
