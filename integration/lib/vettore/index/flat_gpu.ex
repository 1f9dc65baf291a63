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

  def put_many(%Collection{index_state: ref, dimensions: d}, embeddings) do
    # one binary instead of count * d list cells (the reference's marshalling cost, SURVEY 8a17)
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

  @doc "collection.ex:276-295 on the resident corpus: no ETS scan, no per-record NIF traffic."
  def quantized_search(%Collection{} = collection, query, opts \\ []) do
    limit = Keyword.get(opts, :limit, 10)
    candidates = Keyword.get(opts, :candidates, max(limit * 10, limit))

    with :ok <- validate_limit(limit),
         :ok <- validate_limit(candidates),
         {:ok, prepared} <- Collection.prepare_query(collection, query),
         {:ok, hits} <- Nifs.flat_quantized_search(collection.index_state, prepared, candidates, limit) do
      {:ok, Enum.flat_map(hits, &to_result(collection, &1))}
    end
  end

  @doc "collection.ex:245-260 on the resident corpus."
  def funnel_search(%Collection{dimensions: d} = collection, query, opts \\ []) do
    limit = Keyword.get(opts, :limit, 10)
    candidates = Keyword.get(opts, :candidates, max(limit * 10, limit))
    stages = Keyword.get(opts, :stages, [Keyword.get(opts, :dimensions, min(d, 128))])

    with :ok <- validate_limit(limit),
         :ok <- validate_limit(candidates),
         true <- Enum.all?(stages, &(is_integer(&1) and &1 > 0 and &1 <= d)) or {:error, :invalid_stages},
         {:ok, prepared} <- Collection.prepare_query(collection, query),
         {:ok, hits} <- Nifs.flat_funnel_search(collection.index_state, prepared, stages, candidates, limit) do
      {:ok, Enum.flat_map(hits, &to_result(collection, &1))}
    end
  end

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
end
