defmodule Vettore.Index.FlatGpu do
  @moduledoc """
  `Vettore.Index` implementation (lib/vettore/index.ex:12-17) backed by an MI355X.

      Vettore.new(dimensions: 768, metric: :cosine, index: Vettore.Index.FlatGpu,
                  index_options: [device: 0])
      # or the whole node behind one collection:
      Vettore.new(dimensions: 768, metric: :l2, index: Vettore.Index.FlatGpu,
                  index_options: [devices: Enum.to_list(0..7)])

  Same steps as `Vettore.Index.Flat` (lib/vettore/index/flat.ex:29-57, :72-112): the
  collection has already validated and normalised embeddings on `put*`
  (collection.ex:921-937); `search/3` prepares the query itself (flat.ex:53) and drops
  hits whose id is no longer in ETS (flat.ex:88-89).  Results are identical to the
  built-in flat index: ids, order (rank, then id bytes) and raw scores.
  """
  @behaviour Vettore.Index

  alias Vettore.{Collection, Distance, Embedding, Result}
  alias Vettore.Gpu.Nifs

  @max_nif_usize 4_294_967_295
  @codes %{l2: 0, l2_squared: 1, cosine: 2, inner_product: 3, negative_inner_product: 4,
           manhattan: 5, chebyshev: 6, hamming: 7, jaccard: 8}

  @impl true
  def new(metric, options) when is_list(options) do
    with true <- Keyword.keyword?(options),
         {:ok, devices} <- devices(options),
         {:ok, code} <- Map.fetch(@codes, metric) do
      # the NIF hands back a bare reference, or {:error, message} (no such device, out of memory)
      case Nifs.flat_new(code, devices) do
        {:error, _} = error -> error
        ref -> {:ok, ref}
      end
    else
      :error -> {:error, {:unsupported_flat_metric, metric}}
      _ -> {:error, :invalid_flat_options}
    end
  end

  def new(_metric, _options), do: {:error, :invalid_flat_options}

  @impl true
  def put(%Collection{index_state: ref}, %Embedding{id: id, vector: vector}),
    do: unit(Nifs.flat_insert(ref, id, vector))

  @impl true
  def put_many(%Collection{}, []), do: :ok

  # a handful of rows as the reference sends them (flat.ex:35-39: [{id, vector}]) ...
  def put_many(%Collection{index_state: ref}, embeddings) when length(embeddings) < 8 do
    unit(Nifs.flat_insert_many(ref, Enum.map(embeddings, fn %Embedding{id: id, vector: v} -> {id, v} end)))
  end

  # ... a real batch as one binary instead of count * d list cells (the reference's marshalling cost, SURVEY 8a17)
  def put_many(%Collection{index_state: ref, dimensions: d}, embeddings) do
    ids = Enum.map(embeddings, & &1.id)
    rows = for %Embedding{vector: v} <- embeddings, x <- v, into: <<>>, do: <<x::float-32-native>>
    unit(Nifs.flat_load_binary(ref, ids, rows, d))
  end

  @impl true
  def delete(%Collection{index_state: ref}, id), do: unit(Nifs.flat_delete(ref, id))

  @impl true
  def search(%Collection{} = collection, query, opts) do
    with :ok <- validate_opts(opts),
         limit = Keyword.get(opts, :limit, 10),
         :ok <- validate_limit(limit),
         {:ok, prepared} <- Collection.prepare_query(collection, query),
         {:ok, hits} <- Nifs.flat_search(collection.index_state, prepared, limit) do
      {:ok, Enum.flat_map(hits, &to_result(collection, &1))}
    end
  end

  @doc "B queries in one pass over the corpus (FP32 matrix cores + exact rescoring)."
  def search_batch(%Collection{} = collection, queries, opts \\ []) do
    with :ok <- validate_opts(opts),
         limit = Keyword.get(opts, :limit, 10),
         :ok <- validate_limit(limit),
         {:ok, prepared} <- prepare_all(collection, queries),
         {:ok, lists} <- Nifs.flat_search_batch(collection.index_state, prepared, limit) do
      {:ok, Enum.map(lists, fn hits -> Enum.flat_map(hits, &to_result(collection, &1)) end)}
    end
  end

  # ---- the collection's staged searches on the resident corpus ------------------------------------
  # `Vettore.quantized_search/3`, `funnel_search/3` and `hybrid_search/3` (collection.ex:263-345) never ask the index
  # module: they read every record out of ETS and call the CPU NIFs.  With the three-line dispatch of INTEGRATION.md
  # section 3 in collection.ex they land here instead -- same arguments (the collection has checked the option KEYS,
  # collection.ex:237, :267, :330), same validation order, same results; without it they keep working on a FlatGpu
  # collection at the reference's own speed.

  @doc "collection.ex:276-295 on the resident corpus: no ETS scan, no per-record NIF traffic."
  def quantized_search(%Collection{} = collection, query, opts \\ []) do
    limit = Keyword.get(opts, :limit, 10)
    candidates = Keyword.get(opts, :candidates, max_candidates(limit))

    with :ok <- validate_limit(limit),
         :ok <- validate_candidates(candidates, limit),
         {:ok, prepared} <- Collection.prepare_query(collection, query),
         {:ok, hits} <- Nifs.flat_quantized_search(collection.index_state, prepared, candidates, limit) do
      {:ok, Enum.flat_map(hits, &to_result(collection, &1))}
    end
  end

  @doc "The same for a list of queries: groups of eight share one sweep of the sign bits."
  def quantized_search_batch(%Collection{} = collection, queries, opts \\ []) do
    limit = Keyword.get(opts, :limit, 10)
    candidates = Keyword.get(opts, :candidates, max_candidates(limit))

    with :ok <- validate_limit(limit),
         :ok <- validate_candidates(candidates, limit),
         {:ok, prepared} <- prepare_all(collection, queries),
         {:ok, lists} <- Nifs.flat_quantized_search_batch(collection.index_state, prepared, candidates, limit) do
      {:ok, Enum.map(lists, fn hits -> Enum.flat_map(hits, &to_result(collection, &1)) end)}
    end
  end

  @doc "collection.ex:245-260 on the resident corpus."
  def funnel_search(%Collection{dimensions: d} = collection, query, opts \\ []) do
    limit = Keyword.get(opts, :limit, 10)
    candidates = Keyword.get(opts, :candidates, max_candidates(limit))
    stages = funnel_stages(d, opts)

    with :ok <- validate_limit(limit),
         :ok <- validate_candidates(candidates, limit),
         :ok <- validate_stages(stages, d),
         {:ok, prepared} <- Collection.prepare_query(collection, query),
         {:ok, hits} <- Nifs.flat_funnel_search(collection.index_state, prepared, stages, candidates, limit) do
      {:ok, Enum.flat_map(hits, &to_result(collection, &1))}
    end
  end

  @doc "The same for a list of queries with one set of stages: groups of eight share the sweep of the prefixes."
  def funnel_search_batch(%Collection{dimensions: d} = collection, queries, opts \\ []) do
    limit = Keyword.get(opts, :limit, 10)
    candidates = Keyword.get(opts, :candidates, max_candidates(limit))
    stages = funnel_stages(d, opts)

    with :ok <- validate_limit(limit),
         :ok <- validate_candidates(candidates, limit),
         :ok <- validate_stages(stages, d),
         {:ok, prepared} <- prepare_all(collection, queries),
         {:ok, lists} <- Nifs.flat_funnel_search_batch(collection.index_state, prepared, stages, candidates, limit) do
      {:ok, Enum.map(lists, fn hits -> Enum.flat_map(hits, &to_result(collection, &1)) end)}
    end
  end

  @doc """
  collection.ex:325-345 with `rerank: :exact`: the generators' candidate sets (`:funnel`, `:quantized`, `:search`,
  each `name` or `{name, opts}`, collection.ex:515-592), their union, the exact rerank -- on the resident corpus.
  A `{:multi_vector, _}` rerank is not on the flat path: `{:error, {:invalid_rerank, _}}`, as for any unknown one.
  """
  def hybrid_search(%Collection{dimensions: d} = collection, query, opts \\ []) do
    limit = Keyword.get(opts, :limit, 10)
    generators = Keyword.get(opts, :generators, [:funnel, :quantized])
    rerank = Keyword.get(opts, :rerank, :exact)

    with :ok <- validate_limit(limit),
         {:ok, prepared} <- Collection.prepare_query(collection, query),
         {:ok, spec} <- generator_spec(generators, d, limit),
         :ok <- if(rerank == :exact, do: :ok, else: {:error, {:invalid_rerank, rerank}}),
         {:ok, hits} <- Nifs.flat_hybrid_search(collection.index_state, prepared, spec, limit) do
      {:ok, Enum.flat_map(hits, &to_result(collection, &1))}
    end
  end

  # The stateless helpers of Vettore.Nifs (nifs.rs:107-129, :151-175) on the device, for callers that hold their own
  # candidate lists: same terms in and out.
  defdelegate normalize_l2(vector), to: Nifs
  defdelegate compress_sign_bits(vector), to: Nifs
  defdelegate vector_top_k(vectors, query, metric_code, dimensions, limit), to: Nifs
  defdelegate binary_top_k(vectors, query, dimensions, limit), to: Nifs

  # collection.ex:510 / :547
  defp max_candidates(limit) when is_integer(limit), do: max(limit * 10, limit)
  defp max_candidates(_), do: 0

  # collection.ex:660-672
  defp funnel_stages(d, opts) do
    cond do
      Keyword.has_key?(opts, :stages) -> Keyword.fetch!(opts, :stages)
      Keyword.has_key?(opts, :dimensions) -> [Keyword.fetch!(opts, :dimensions)]
      true -> [min(d, 128)]
    end
  end

  # collection.ex:905-913
  defp validate_stages(stages, d) when is_list(stages) and stages != [] do
    if Enum.all?(stages, &(is_integer(&1) and &1 > 0 and &1 <= d)), do: :ok, else: {:error, :invalid_stages}
  end

  defp validate_stages(_stages, _d), do: {:error, :invalid_stages}

  # [{kind, candidates, stages}] as the NIF takes them (kind 0 funnel, 1 quantized, 2 search); errors as
  # run_hybrid_generator's (collection.ex:536-556, :1136-1142)
  defp generator_spec(generators, d, limit) when is_list(generators) and generators != [] do
    Enum.reduce_while(generators, {:ok, []}, fn generator, {:ok, acc} ->
      case one_generator(generator, d, limit) do
        {:ok, entry} -> {:cont, {:ok, [entry | acc]}}
        error -> {:halt, error}
      end
    end)
    |> case do
      {:ok, acc} -> {:ok, Enum.reverse(acc)}
      error -> error
    end
  end

  defp generator_spec(_generators, _d, _limit), do: {:error, :invalid_generators}

  defp one_generator(name, d, limit) when is_atom(name), do: one_generator({name, []}, d, limit)

  defp one_generator({name, opts}, d, limit) when is_atom(name) and is_list(opts) do
    allowed = if name == :funnel, do: [:candidates, :stages, :dimensions], else: [:candidates]
    candidates = Keyword.get(opts, :candidates, max_candidates(limit))

    cond do
      name not in [:funnel, :quantized, :search, :hnsw] -> {:error, {:unknown_generator, name}}
      not Keyword.keyword?(opts) -> {:error, :invalid_options}
      (dup = Enum.find(Keyword.keys(opts), &(length(Keyword.get_values(opts, &1)) > 1))) != nil -> {:error, {:duplicate_option, dup}}
      (extra = Keyword.keys(opts) -- allowed) != [] -> {:error, {:unsupported_option, hd(extra)}}
      name == :hnsw -> {:error, :hnsw_index_required}
      not (is_integer(candidates) and candidates > 0 and candidates <= @max_nif_usize) -> {:error, :invalid_candidates}
      name == :funnel ->
        stages = funnel_stages(d, opts)
        with :ok <- validate_stages(stages, d), do: {:ok, {0, candidates, stages}}
      name == :quantized -> {:ok, {1, candidates, []}}
      true -> {:ok, {2, candidates, []}}
    end
  end

  defp one_generator(generator, _d, _limit), do: {:error, {:invalid_generator, generator}}

  # `index_options: [device: 0]` (default) or `[devices: [0, 1, 2, 3, 4, 5, 6, 7]]`: one resource whose
  # rows are spread over the GPUs of the node (collection.ex:100-103 hands the options over verbatim).
  defp devices(options) do
    case options do
      [] -> {:ok, [0]}
      [device: d] when is_integer(d) and d >= 0 -> {:ok, [d]}
      [devices: [_ | _] = ds] -> if Enum.all?(ds, &(is_integer(&1) and &1 >= 0)), do: {:ok, ds}, else: :invalid
      _ -> :invalid
    end
  end

  # -- helpers, as in lib/vettore/index/flat.ex:72-112 -----------------------------
  defp prepare_all(collection, queries) do
    Enum.reduce_while(queries, {:ok, []}, fn q, {:ok, acc} ->
      case Collection.prepare_query(collection, q) do
        {:ok, p} -> {:cont, {:ok, [p | acc]}}
        error -> {:halt, error}
      end
    end)
    |> case do
      {:ok, acc} -> {:ok, Enum.reverse(acc)}
      error -> error
    end
  end

  # a hit whose record has left ETS in the meantime is dropped, like flat.ex does
  defp to_result(%Collection{metric: metric, score: mode} = collection, {id, raw}) do
    with {:ok, %Embedding{value: value, metadata: metadata}} <- Collection.get(collection, id) do
      {score, distance} = Distance.result_values(metric, raw, mode)
      [%Result{id: id, value: value, score: score, distance: distance, metric: metric, metadata: metadata}]
    else
      _ -> []
    end
  end

  defp unit({:ok, {}}), do: :ok
  defp unit({:error, _} = error), do: error

  defp validate_opts(opts) do
    only_limit? = is_list(opts) and Keyword.keyword?(opts) and Keyword.keys(opts) -- [:limit] == []
    if only_limit?, do: :ok, else: {:error, :invalid_search_options}
  end

  defp validate_limit(limit) when is_integer(limit) and limit > 0 and limit <= @max_nif_usize, do: :ok
  defp validate_limit(_), do: {:error, :invalid_limit}

  # collection.ex:889-895
  defp validate_candidates(candidates, limit)
       when is_integer(candidates) and candidates >= limit and candidates > 0 and candidates <= @max_nif_usize,
       do: :ok

  defp validate_candidates(_candidates, _limit), do: {:error, :invalid_candidates}
end
